// scene_loader.h -- host-side scene loader above the C ABI (include/rtcore.h).
//
// Same interface and file grammar as upstream's vrt::SceneLoader (include/sceneloader.h:18-43,
// csrc/sceneloader.cpp:39-401): load_scene(filename) reads the scene text, loads every object's OBJ/MTL,
// assigns material ids, applies the model transform and builds one BVH per shape; the results stay on the
// host in the layout rt_upload_scene takes, and upload(ctx) hands them to a device context.
#ifndef RT_SCENE_LOADER_H
#define RT_SCENE_LOADER_H

#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "rtcore.h"

namespace rth {

struct ObjModel;

// include/common.cuh:19-33
struct RenderConfig {
    int max_depth;
    int max_surface;
    int width;
    int height;
    int n_samples;
    int batch_size;
    float fov;               // radians
    float gamma;
    float russian_roulette;
    rt_float3 background;
    rt_float3 camera_pos;
    rt_float3 camera_dir;
    rt_float3 camera_up;
};

struct HostTexture {
    int width = 0, height = 0;
    std::vector<uint8_t> rgba;       // row 0 = bottom row of the image file (csrc/sceneloader.cpp:229-230)
};

struct HostObject {                  // one per OBJ shape with at least one triangle (csrc/sceneloader.cpp:132-186)
    std::vector<rt_triangle> triangles;
    std::vector<rt_bvh_node> bvh;
    rt_transform transform;
};

class SceneLoader {
    std::vector<HostObject> objects_;
    std::vector<rt_material> materials_;
    std::vector<HostTexture> textures_;
    RenderConfig config_;
    bool device_tree_ = false;       // RT_GPU_BVH=1: the scene tree is built on the device at upload
    // model files already read during this load_scene, by path and MTL directory: a scene that places one model many
    // times (upstream reads the file again for every object block) reads it once
    std::map<std::string, std::shared_ptr<ObjModel>> parsed_;

    int load_object(std::istream& file);
    int load_texture(const std::string& filename, rt_material& material);

public:
    SceneLoader();

    // 0 on success, -1 on any error (message on stderr), as upstream.
    int load_scene(const std::string& filename);

    // Hands the loaded scene to a device context (rt_upload_scene); 0 or a negative rt_status.
    int upload(rt_context* ctx) const;

    int num_object() const { return int(objects_.size()); }
    int num_material() const { return int(materials_.size()); }
    const std::vector<HostObject>& objects() const { return objects_; }
    const std::vector<rt_material>& materials() const { return materials_; }
    const std::vector<HostTexture>& textures() const { return textures_; }

    RenderConfig& config() { return config_; }
    const RenderConfig& config() const { return config_; }
};

}  // namespace rth
#endif
