// scene_loader.cpp -- see scene_loader.h.  Grammar and rules follow csrc/sceneloader.cpp (cited per block).
#include "scene_loader.h"

#include "file_util.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>

#include "image_io.h"
#include "obj_model.h"

namespace rth {

namespace {

const char* kMaterialTypeName[] = {"NONE", "LIGHT", "LAMBERTIAN", "METAL", "REFRACTIVE", "REFLECTIVE"};

// Whitespace-separated words, as `stream >> std::string` delivers them; false at the end of the file.
bool next_word(std::istream& in, std::string& word) {
    word.clear();
    return bool(in >> word);
}

std::string rest_of_line(std::istream& in) {
    std::string line;
    std::getline(in, line);
    if (!line.empty() && line.back() == '\r') line.pop_back();
    return line;
}

// A number must be a whole word.  (Upstream's `stream >> float` would leave the stream failed on anything else
// and then spin forever in `while (!file.eof())`; here that is an error.)
bool read_float(std::istream& in, float& out, const char* what) {
    std::string w;
    if (!next_word(in, w)) { std::cerr << "Error: missing number after " << what << std::endl; return false; }
    char* end = nullptr;
    const float v = std::strtof(w.c_str(), &end);
    if (end == w.c_str() || *end != '\0') { std::cerr << "Error: bad number '" << w << "' after " << what << std::endl; return false; }
    out = v;
    return true;
}

bool read_int(std::istream& in, int& out, const char* what) {
    std::string w;
    if (!next_word(in, w)) { std::cerr << "Error: missing number after " << what << std::endl; return false; }
    char* end = nullptr;
    const long v = std::strtol(w.c_str(), &end, 10);
    if (end == w.c_str() || *end != '\0') { std::cerr << "Error: bad integer '" << w << "' after " << what << std::endl; return false; }
    out = int(v);
    return true;
}

bool read_float3(std::istream& in, rt_float3& out, const char* what) {
    return read_float(in, out.x, what) && read_float(in, out.y, what) && read_float(in, out.z, what);
}

// float3 helpers with the operand order of include/common.cuh:75-94
rt_float3 sub3(const rt_float3& a, const rt_float3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
rt_float3 cross3(const rt_float3& a, const rt_float3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
rt_float3 normalize3(const rt_float3& a) {
    const float norm = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x / norm, a.y / norm, a.z / norm};
}

rt_material make_material(int type, rt_float3 albedo, float optical_density, float metal_fuzz) {   // include/material.cuh:28-40
    rt_material m;
    std::memset(&m, 0, sizeof m);
    m.type = type;
    m.albedo = albedo;
    m.optical_density = optical_density;
    m.metal_fuzz = metal_fuzz;
    m.texture = -1;
    return m;
}

}  // namespace

SceneLoader::SceneLoader() { std::memset(&config_, 0, sizeof config_); }

// csrc/sceneloader.cpp:39-221
int SceneLoader::load_object(std::istream& file) {
    rt_transform transform;
    transform.translation = {0.0f, 0.0f, 0.0f};
    transform.rotation = {0.0f, 0.0f, 0.0f};
    transform.scale = {1.0f, 1.0f, 1.0f};
    int type = RT_MAT_NONE;
    rt_float3 albedo = {0.0f, 0.0f, 0.0f};
    float optical_density = 1.0f;
    float metal_fuzz = 0.0f;
    bool disable_normal = false;
    bool simple_material = false;
    std::string objfilename, mtldir, key;

    // keys of an object block (:57-96); the block ends at `end` or at the end of the file
    while (next_word(file, key)) {
        if (key == "position") { if (!read_float3(file, transform.translation, "position")) return -1; }
        else if (key == "rotation") { if (!read_float3(file, transform.rotation, "rotation")) return -1; }
        else if (key == "scale") { if (!read_float3(file, transform.scale, "scale")) return -1; }
        else if (key == "path" || key == "mtldir") {
            // the rest of the line after one separating blank: file names may contain blanks (:62-69)
            const std::string rest = rest_of_line(file);
            if (rest.size() < 1) { std::cerr << "Error: " << key << " needs a value" << std::endl; return -1; }
            (key == "path" ? objfilename : mtldir) = rest.substr(1);
        }
        else if (key == "albedo") { if (!read_float3(file, albedo, "albedo")) return -1; }
        else if (key == "optical_density") { if (!read_float(file, optical_density, "optical_density")) return -1; }
        else if (key == "metal_fuzz") { if (!read_float(file, metal_fuzz, "metal_fuzz")) return -1; }
        else if (key == "end") { break; }
        else if (key == "disable_normal") { disable_normal = true; }
        else if (key == "simple_material") { simple_material = true; }
        else if (key == "#") { rest_of_line(file); }              // a comment is the word "#" and the rest of its line
        else if (key == "type") {
            std::string t;
            next_word(file, t);
            if (t == "light") type = RT_MAT_LIGHT;
            else if (t == "lambertian") type = RT_MAT_LAMBERTIAN;
            else if (t == "metal") type = RT_MAT_METAL;
            else if (t == "refractive") type = RT_MAT_REFRACTIVE;
            else if (t == "reflective") type = RT_MAT_REFLECTIVE;
            else {
                std::cerr << "Warning: Unknown material type " << t << std::endl;
                return -1;
            }
        }
        else {
            std::cerr << "Warning: Unknown key in object " << key << std::endl;
            return -1;
        }
    }
    if (objfilename.empty()) {
        std::cerr << "Error: No object file specified" << std::endl;
        return -1;
    }
    if (type == RT_MAT_NONE) {
        std::cerr << "Error: No material type specified" << std::endl;
        return -1;
    }

    std::cout << "Loading object " << objfilename << std::endl;
    if (simple_material) std::cout << "!   Simple Material type: " << kMaterialTypeName[type] << std::endl;
    std::shared_ptr<ObjModel>& parsed = parsed_[objfilename + "\n" + mtldir];
    if (!parsed) {
        parsed = std::make_shared<ObjModel>();
        if (!load_obj(objfilename, mtldir, *parsed)) {            // :110-119
            if (!parsed->error.empty()) std::cerr << "Error: " << parsed->error << std::endl;
            parsed.reset();
            return -1;
        }
        if (!parsed->warning.empty()) std::cerr << "Warning: " << parsed->warning << std::endl;
    }
    const ObjModel& model = *parsed;
    const int material_offset = int(materials_.size());
    if (model.shapes.empty()) {
        std::cerr << "Error: No shapes in object file" << std::endl;
        return -1;
    }
    const size_t n_pos = model.vertices.size() / 3, n_nrm = model.normals.size() / 3, n_uv = model.texcoords.size() / 2;

    // one render object per shape that has triangles (:132-186)
    for (const ObjShape& shape : model.shapes) {
        std::cout << "    Loading shape " << shape.name << "...";
        const size_t n_triangles = shape.indices.size() / 3;
        if (n_triangles == 0) {
            std::cerr << "\n! Warning: Shape " << shape.name << " has no triangles, ignored." << std::endl;
            continue;
        }
        objects_.emplace_back();
        HostObject& obj = objects_.back();
        obj.transform = transform;
        obj.triangles.resize(n_triangles);
        std::memset(obj.triangles.data(), 0, n_triangles * sizeof(rt_triangle));      // texcoords upstream leaves uninitialised are 0 here
        for (size_t t = 0; t < n_triangles; t++) {
            rt_triangle& tri = obj.triangles[t];
            const ObjIndex* ix = &shape.indices[3 * t];
            for (int j = 0; j < 3; j++) {
                if (ix[j].v < 0 || size_t(ix[j].v) >= n_pos) {
                    std::cerr << "\nError: vertex index out of range in shape " << shape.name << std::endl;   // upstream reads out of bounds here
                    return -1;
                }
                tri.v[j].position = {model.vertices[3 * size_t(ix[j].v)], model.vertices[3 * size_t(ix[j].v) + 1], model.vertices[3 * size_t(ix[j].v) + 2]};
            }
            if (disable_normal || ix[0].vn < 0) {                 // the FIRST corner decides (:157)
                const rt_float3 e1 = sub3(tri.v[1].position, tri.v[0].position);
                const rt_float3 e2 = sub3(tri.v[2].position, tri.v[0].position);
                const rt_float3 n = normalize3(cross3(e1, e2));
                for (int j = 0; j < 3; j++) tri.v[j].normal = n;
            } else {
                for (int j = 0; j < 3; j++) {
                    if (ix[j].vn < 0 || size_t(ix[j].vn) >= n_nrm) {
                        std::cerr << "\nError: normal index out of range in shape " << shape.name << std::endl;
                        return -1;
                    }
                    tri.v[j].normal = {model.normals[3 * size_t(ix[j].vn)], model.normals[3 * size_t(ix[j].vn) + 1], model.normals[3 * size_t(ix[j].vn) + 2]};
                }
            }
            if (ix[0].vt >= 0) {                                  // again the first corner decides (:173)
                for (int j = 0; j < 3; j++) {
                    if (ix[j].vt < 0 || size_t(ix[j].vt) >= n_uv) {
                        std::cerr << "\nError: texture coordinate index out of range in shape " << shape.name << std::endl;
                        return -1;
                    }
                    tri.v[j].texcoord[0] = model.texcoords[2 * size_t(ix[j].vt)];
                    tri.v[j].texcoord[1] = model.texcoords[2 * size_t(ix[j].vt) + 1];
                }
            }
            tri.material_id = material_offset;                    // the object's own material ...
            if (!simple_material && shape.material_ids[t] >= 0) tri.material_id += shape.material_ids[t] + 1;   // ... or the face's MTL record (:180-183)
        }
        std::cout << n_triangles << " triangles" << std::endl;
    }

    // the object's own material first, then one per MTL record (:188-219)
    materials_.push_back(make_material(type, albedo, optical_density, metal_fuzz));
    if (!simple_material && !model.materials.empty()) {
        for (const ObjMaterial& mat : model.materials) {
            switch (mat.illum) {                                  // :192-199; 4 and 5 are upstream's own glass and mirror codes
            case 1: type = RT_MAT_LAMBERTIAN; break;
            case 2: type = RT_MAT_METAL; break;
            case 3: type = RT_MAT_LIGHT; break;
            case 4: type = RT_MAT_REFRACTIVE; break;
            case 5: type = RT_MAT_REFLECTIVE; break;
            default: type = RT_MAT_LAMBERTIAN; break;
            }
            rt_float3 a;
            if (type == RT_MAT_LIGHT) {
                a = {mat.emission[0], mat.emission[1], mat.emission[2]};
            } else {
                a = {mat.diffuse[0], mat.diffuse[1], mat.diffuse[2]};
                if (type == RT_MAT_REFRACTIVE) {
                    optical_density = mat.ior;                    // stays in force for the records that follow (:206)
                    const float clear = 1.0f - mat.dissolve;
                    a = {clear * 1.0f + mat.dissolve * a.x, clear * 1.0f + mat.dissolve * a.y, clear * 1.0f + mat.dissolve * a.z};
                }
            }
            materials_.push_back(make_material(type, a, optical_density, metal_fuzz));
            if (!mat.diffuse_texname.empty()) {
                const std::string texname = mtldir + "/" + mat.diffuse_texname;
                std::cout << "    Loading texture " << texname << std::endl;
                if (load_texture(texname, materials_.back()) < 0) return -1;
            }
        }
    }
    return 0;
}

// csrc/sceneloader.cpp:223-252: decode to 8-bit colour, flip top to bottom, RGBA
int SceneLoader::load_texture(const std::string& filename, rt_material& material) {
    Image8 img;
    std::string error;
    if (!read_image(filename, img, error)) {
        std::cerr << "Error: Cannot load texture " << filename << " (" << error << ")" << std::endl;
        return -1;
    }
    HostTexture tex;
    tex.width = img.width;
    tex.height = img.height;
    tex.rgba.resize(size_t(img.width) * size_t(img.height) * 4);
    for (int y = 0; y < img.height; y++) {
        const uint8_t* src = &img.rgb[size_t(img.height - 1 - y) * size_t(img.width) * 3];
        uint8_t* dst = &tex.rgba[size_t(y) * size_t(img.width) * 4];
        for (int x = 0; x < img.width; x++) {
            dst[4 * x] = src[3 * x]; dst[4 * x + 1] = src[3 * x + 1]; dst[4 * x + 2] = src[3 * x + 2]; dst[4 * x + 3] = 255;
        }
    }
    material.texture = int(textures_.size());
    textures_.push_back(std::move(tex));
    return 0;
}

// csrc/sceneloader.cpp:254-401
int SceneLoader::load_scene(const std::string& filename) {
    std::memset(&config_, 0, sizeof config_);
    objects_.clear();
    materials_.clear();
    textures_.clear();
    parsed_.clear();
    int cur_obj = 0;
    std::ifstream file(filename.c_str());
    if (!file.is_open() || !is_plain_file(filename)) {
        std::cerr << "Error: Cannot open file " << filename << std::endl;
        return -1;
    }
    std::string key;
    next_word(file, key);
    if (key != "scene_config") {
        std::cerr << "Error: Invalid configuration file" << std::endl;
        return -1;
    }
    bool ended = false;
    std::string last_key;
    while (next_word(file, key)) {
        last_key = key;
        if (key == "max_ray_depth") { if (!read_int(file, config_.max_depth, "max_ray_depth")) return -1; }
        else if (key == "max_surface") { if (!read_int(file, config_.max_surface, "max_surface")) return -1; }
        else if (key == "#") { rest_of_line(file); }
        else if (key == "size") { if (!read_int(file, config_.width, "size") || !read_int(file, config_.height, "size")) return -1; }
        else if (key == "sample_per_pixel") { if (!read_int(file, config_.n_samples, "sample_per_pixel")) return -1; }
        else if (key == "fov") {
            if (!read_float(file, config_.fov, "fov")) return -1;
            config_.fov = float(double(config_.fov) * M_PI / double(180.0f));      // degrees in the file (:287)
        }
        else if (key == "gamma") { if (!read_float(file, config_.gamma, "gamma")) return -1; }
        else if (key == "background") { if (!read_float3(file, config_.background, "background")) return -1; }
        else if (key == "camera_pos") { if (!read_float3(file, config_.camera_pos, "camera_pos")) return -1; }
        else if (key == "camera_dir") { if (!read_float3(file, config_.camera_dir, "camera_dir")) return -1; }
        else if (key == "camera_up") { if (!read_float3(file, config_.camera_up, "camera_up")) return -1; }
        else if (key == "russian_roulette") { if (!read_float(file, config_.russian_roulette, "russian_roulette")) return -1; }
        else if (key == "object") {
            const int ret = load_object(file);
            if (ret < 0) return ret;
            cur_obj++;
        }
        else if (key == "end") { ended = true; break; }
        else {
            std::cerr << "Warning: Unknown key " << key << std::endl;
            return -1;
        }
    }
    if (!ended && last_key == "object") {
        // Upstream's loop runs once more on the stale key when the file stops after an object block without the
        // closing `end`, and that phantom object has no path (:269-313, :97-100).
        std::cerr << "Error: No object file specified" << std::endl;
        return -1;
    }
    if (cur_obj == 0) {
        std::cerr << "Error: No object loaded" << std::endl;
        return -1;
    }
    // model transform, then one BVH per object (:336-375); both run on the host here
    int n_bvh_node_total = 0;
    for (size_t i = 0; i < objects_.size(); i++) {
        HostObject& o = objects_[i];
        std::cout << "Copying object\t" << (i + 1) << " /\t" << objects_.size() << '\r' << std::flush;
        if (rt_apply_transform(&o.transform, o.triangles.data(), int(o.triangles.size())) != RT_OK) {
            std::cerr << "\nError: " << rt_last_error() << std::endl;
            return -1;
        }
    }
    parsed_.clear();                                              // the parsed model files are not needed any more
    // RT_GPU_BVH=1: no trees are built here; upload() asks the library to build one on the device (RT_OPT_GPU_BVH)
    const char* on_device = std::getenv("RT_GPU_BVH");
    device_tree_ = on_device && std::atoi(on_device) != 0;
    if (device_tree_) {
        std::cout << "\nBVH tree: left to the device" << std::endl;
        return 0;
    }
    std::cout << "\nBuilding BVH tree... ";
    for (HostObject& o : objects_) {
        o.bvh.resize(2 * o.triangles.size());
        const int n = rt_build_bvh(o.triangles.data(), int(o.triangles.size()), o.bvh.data(), int(o.bvh.size()));
        if (n < 0) {
            std::cerr << "\nError: " << rt_last_error() << std::endl;
            return -1;
        }
        o.bvh.resize(size_t(n));
        n_bvh_node_total += n;
    }
    std::cout << "Done. total " << n_bvh_node_total << " nodes" << std::endl;
    return 0;
}

int SceneLoader::upload(rt_context* ctx) const {
    std::vector<rt_object> objs(objects_.size());
    for (size_t i = 0; i < objects_.size(); i++) {
        objs[i].triangles = objects_[i].triangles.data();
        objs[i].bvh = device_tree_ ? nullptr : objects_[i].bvh.data();
        objs[i].n_triangles = int(objects_[i].triangles.size());
        objs[i].n_bvh_nodes = device_tree_ ? 0 : int(objects_[i].bvh.size());
    }
    std::vector<rt_texture> texs(textures_.size());
    for (size_t i = 0; i < textures_.size(); i++) {
        texs[i].rgba = textures_[i].rgba.data();
        texs[i].width = textures_[i].width;
        texs[i].height = textures_[i].height;
    }
    if (device_tree_ && rt_set_option(ctx, RT_OPT_GPU_BVH, 1) != RT_OK) return RT_ERR_INVALID;
    return rt_upload_scene(ctx, objs.data(), int(objs.size()), materials_.data(), int(materials_.size()), texs.data(), int(texs.size()));
}

}  // namespace rth
