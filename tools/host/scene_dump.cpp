// scene_dump.cpp -- loads a scene file with the product's SceneLoader (host only, no device) and writes what it
// built: a text summary on stdout and, if a directory is given, the raw arrays (triangles, BVH nodes, materials,
// textures) for byte-level comparison.  Usage: scene_dump <scene_file> [out_dir]
#include <cstdio>
#include <fstream>
#include <iostream>

#include "scene_loader.h"

static void write_file(const std::string& path, const void* p, size_t n) {
    std::ofstream f(path.c_str(), std::ios::binary | std::ios::trunc);
    f.write(static_cast<const char*>(p), std::streamsize(n));
}

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: scene_dump <scene_file> [out_dir]\n"; return 2; }
    rth::SceneLoader loader;
    const int rc = loader.load_scene(argv[1]);
    if (rc != 0) { std::printf("load_scene %d\n", rc); return 1; }
    const rth::RenderConfig& c = loader.config();
    std::printf("config depth %d surface %d size %d %d spp %d fov %a gamma %a rr %a\n", c.max_depth, c.max_surface, c.width, c.height, c.n_samples,
                c.fov, c.gamma, c.russian_roulette);
    std::printf("background %a %a %a pos %a %a %a dir %a %a %a up %a %a %a\n", c.background.x, c.background.y, c.background.z, c.camera_pos.x,
                c.camera_pos.y, c.camera_pos.z, c.camera_dir.x, c.camera_dir.y, c.camera_dir.z, c.camera_up.x, c.camera_up.y, c.camera_up.z);
    std::printf("objects %d materials %d textures %zu\n", loader.num_object(), loader.num_material(), loader.textures().size());
    for (int i = 0; i < loader.num_object(); i++)
        std::printf("object %d triangles %zu nodes %zu\n", i, loader.objects()[size_t(i)].triangles.size(), loader.objects()[size_t(i)].bvh.size());
    for (int i = 0; i < loader.num_material(); i++) {
        const rt_material& m = loader.materials()[size_t(i)];
        std::printf("material %d type %d albedo %a %a %a od %a fuzz %a tex %d\n", i, m.type, m.albedo.x, m.albedo.y, m.albedo.z, m.optical_density,
                    m.metal_fuzz, m.texture);
    }
    if (argc > 2) {
        const std::string dir = argv[2];
        for (int i = 0; i < loader.num_object(); i++) {
            const rth::HostObject& o = loader.objects()[size_t(i)];
            write_file(dir + "/tri" + std::to_string(i) + ".bin", o.triangles.data(), o.triangles.size() * sizeof(rt_triangle));
            write_file(dir + "/bvh" + std::to_string(i) + ".bin", o.bvh.data(), o.bvh.size() * sizeof(rt_bvh_node));
        }
        write_file(dir + "/materials.bin", loader.materials().data(), loader.materials().size() * sizeof(rt_material));
        for (size_t i = 0; i < loader.textures().size(); i++) {
            const rth::HostTexture& t = loader.textures()[i];
            std::printf("texture %zu size %d %d\n", i, t.width, t.height);
            write_file(dir + "/tex" + std::to_string(i) + ".bin", t.rgba.data(), t.rgba.size());
        }
    }
    return 0;
}
