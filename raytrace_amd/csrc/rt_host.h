// rt_host.h -- host-side helpers behind the C ABI (no HIP types here).
#ifndef RT_HOST_H
#define RT_HOST_H

#include <string>
#include <vector>

#include "../../include/rtcore.h"
#include "rt_packed.h"

namespace rtc {

struct CameraFrame {          // what cuda/camera.cu:49-59 passes to its kernel
    float pos[3];
    float lower_left[3];
    float dx[3];
    float dy[3];
    int width, height;
};

struct PackedScene {
    std::vector<ChildRec> top;       // one per upstream object, in object order
    ChildRec root;                   // the whole scene: root of the object-level tree
    std::vector<Node64> nodes;       // binary nodes (intermediate; not uploaded)
    std::vector<Node4> nodes4;       // two-level nodes the kernels walk; root.ref indexes this array
    std::vector<TriIsect> isect;
    std::vector<TriShade> shade;
    std::vector<TriOrigin> origin;
    int max_depth = 0;               // deepest chain of inner nodes below the scene root (= stack slots needed)
    uint32_t last_tri = REF_NONE;    // packed index of the triangle a walk that enters every box tests last
};

int build_bvh(const rt_triangle* tris, int n, std::vector<rt_bvh_node>& out);
void apply_transform(const rt_transform& xf, rt_triangle* tris, int n);
CameraFrame camera_frame(int width, int height, const float pos[3], const float dir[3], const float up[3], float fov);
int pack_scene(const rt_object* objects, int n_objects, PackedScene& ps, std::string& err);

}  // namespace rtc
#endif
