// rt_build.h -- the scene tree built on the device (rt_build.hip), as rt_upload_scene takes it over.
#ifndef RT_BUILD_H
#define RT_BUILD_H

#include <string>

#include "rt_host.h"

namespace rtc {

struct DeviceScene {
    void* nodes4 = nullptr;      // Node4[n_nodes4] (at least one record allocated)
    void* isect = nullptr;       // TriIsect[n_triangles + 1]
    void* shade = nullptr;       // TriShade[n_triangles]
    void* origin = nullptr;      // TriOrigin[n_triangles]
    ChildRec root;               // box and reference of the whole scene
    uint32_t last_tri = REF_NONE;
    uint64_t n_triangles = 0, n_nodes4 = 0, bytes = 0;
    int stack_need = 0;          // stack entries the walk needs on this tree
    int binary_depth = 0;        // levels of the binary hierarchy the records were made from
    float copy_ms = 0.0f;        // host -> device copy of the caller's triangle arrays
    float build_ms = 0.0f;       // everything after that, on the device
};

// Builds on the current device from the objects' host triangle arrays (their bvh arrays are not read).
// 0 on success; -1 and a message otherwise (nothing stays allocated).
int build_scene_on_device(const rt_object* objects, int n_objects, DeviceScene& out, std::string& err);
void release_device_scene(DeviceScene& scene);

}  // namespace rtc
#endif
