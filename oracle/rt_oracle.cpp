// rt_oracle.cpp -- CPU restatement of the s7a9/RayTrace path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
// calls this file; it exists so that tests, the smoke check and the bench's
// cpu_baseline leg have an independent checker for the HIP kernels.
//
// PARITY UNPINNED: the reference repository holds no tests, golden vectors or
// reproducible outputs for this path (its RNG seed is time(0)), and its own
// sources need the CUDA toolkit headers (cuda.h, cuda_runtime.h,
// curand_kernel.h, thrust) and nvcc, none of which exist in this image, so it
// cannot be compiled here.  This file therefore restates the reference's
// algorithm by hand, function by function, citing the lines it follows.  Two
// things are replaced by specification rather than restated, because they live
// in NVIDIA libraries outside the reference tree:
//   * cuRAND XORWOW (curand_kernel.h, CUDA toolkit version unpinned by the
//     reference's makefile) -> Philox4x32-10 (Salmon et al., SC'11; Random123
//     v1.14 known-answer vectors) keyed by (seed, pixel, sample); the
//     uint32 -> (0,1] mapping is cuRAND's published `x * 2^-32 + 2^-33`.
//   * the CUDA texture unit -> the bilinear formula of the CUDA C Programming
//     Guide, appendix "Texture Fetching" (8-bit fractional weights, border = 0).
//
// Floating-point contract (the parity spec): IEEE-754 binary32, round to
// nearest even, NO fused multiply-add contraction (build with
// -ffp-contract=off), correctly rounded divide and sqrt, subnormals kept.
// Every expression below keeps the reference's operand order.
//
// Reference file:line citations are relative to the upstream repository root.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

// ----------------------------------------------------------------------------
// float3 arithmetic -- include/common.cuh:43-161
// ----------------------------------------------------------------------------
struct f3 { float x, y, z; };
struct f2 { float x, y; } __attribute__((aligned(8)));   // CUDA float2 is 8-aligned

static inline f3 mk(float x, float y, float z) { return f3{x, y, z}; }
// common.cuh:43-46  operator+=
static inline f3 add(const f3& a, const f3& b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
// common.cuh:111-124 operator- is `a += (-b.x,-b.y,-b.z)`; x + (-y) == x - y in IEEE
static inline f3 sub(const f3& a, const f3& b) { return mk(a.x + (-b.x), a.y + (-b.y), a.z + (-b.z)); }
// common.cuh:48-51, 132-146  float3 * float (either side)
static inline f3 mulf(const f3& a, float b) { return mk(a.x * b, a.y * b, a.z * b); }
// common.cuh:53-56  component-wise product
static inline f3 mul3(const f3& a, const f3& b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
// common.cuh:126-130 unary minus is -1.0f * a
static inline f3 neg(const f3& a) { return mulf(a, -1.0f); }
// common.cuh:148-161 a / b is a * (1.f / b)
static inline f3 divf(const f3& a, float b) { return mulf(a, 1.f / b); }
// common.cuh:75-78
static inline f3 normalize(const f3& a) {
    float norm = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return mk(a.x / norm, a.y / norm, a.z / norm);
}
// common.cuh:80-82
static inline float dot3(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// common.cuh:88-94
static inline f3 cross(const f3& a, const f3& b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float get(const f3& a, int i) { return (&a.x)[i]; }   // common.cuh:97-101

// ----------------------------------------------------------------------------
// POD types in the reference's layout
// ----------------------------------------------------------------------------
struct Ray { f3 origin, direction; };                         // primitives.cuh:8-20 (24 B)
struct Vertex { f3 position; f3 normal; f2 texcoord; };       // primitives.cuh:22-26 (32 B)
struct Triangle { int material_id; Vertex v[3]; };            // primitives.cuh:28-31 (104 B)
struct AABB { f3 min, max; };                                 // bvh.h:13-16
struct BVHNode { bool is_leaf; AABB box; int left, right; };  // bvh.h:62-65 (36 B)
static_assert(sizeof(Ray) == 24, "Ray layout");
static_assert(sizeof(Vertex) == 32, "Vertex layout");
static_assert(sizeof(Triangle) == 104, "Triangle layout");
static_assert(sizeof(BVHNode) == 36, "BVHNode layout");

enum MaterialType : uint8_t { NONE, LIGHT, LAMBERTIAN, METAL, REFRACTIVE, REFLECTIVE };  // material.cuh:11-13
// material.cuh:10-26 with the CUDA array/texture handles replaced by an index
// into the scene's texture table (-1 == `cu_array == nullptr`, i.e. untextured).
struct Material {
    int32_t type;
    f3 albedo;
    float optical_density;
    float metal_fuzz;
    int32_t texture;
    int32_t pad;
};
static_assert(sizeof(Material) == 32, "Material layout");

struct Texture { const uint8_t* rgba; int width, height; };   // sceneloader.cpp:223-252: RGBA8, rows already flipped

struct Object { Triangle* triangles; BVHNode* bvh; int n_triangles, n_bvh_nodes; };  // renderobject.cuh:10-15

struct Scene {
    std::vector<Object> objects;
    std::vector<Material> materials;
    std::vector<Texture> textures;
};

// ----------------------------------------------------------------------------
// BVH build -- csrc/bvh.cpp:29-98, include/bvh.h:17-36
// ----------------------------------------------------------------------------
static inline AABB empty_box() {   // bvh.h:17
    return AABB{mk(INFINITY, INFINITY, INFINITY), mk(-INFINITY, -INFINITY, -INFINITY)};
}
static inline void expand(AABB& b, const f3& p) {   // bvh.h:28-31
    b.min = mk(fminf(b.min.x, p.x), fminf(b.min.y, p.y), fminf(b.min.z, p.z));
    b.max = mk(fmaxf(b.max.x, p.x), fmaxf(b.max.y, p.y), fmaxf(b.max.z, p.z));
}
static inline void expand(AABB& b, const AABB& o) {   // bvh.h:33-36
    b.min = mk(fminf(b.min.x, o.min.x), fminf(b.min.y, o.min.y), fminf(b.min.z, o.min.z));
    b.max = mk(fmaxf(b.max.x, o.max.x), fmaxf(b.max.y, o.max.y), fmaxf(b.max.z, o.max.z));
}
static inline AABB tri_box(const Triangle& t) {   // bvh.h:21-26
    AABB b{t.v[0].position, t.v[0].position};
    for (int i = 1; i < 3; i++) expand(b, t.v[i].position);
    return b;
}
static inline void expand_tri(AABB& b, const Triangle& t) {   // bvh.cpp:39 -> AABB(tri) then expand(AABB)
    AABB o = tri_box(t);
    expand(b, o);
}

struct Centroid { f3 position; int index; };   // bvh.cpp:5-8

static void build_recursive(std::vector<Centroid>& c, int start, int end,
                            std::vector<BVHNode>& bvh, const Triangle* tris) {
    if (start + 1 >= end) {                      // bvh.cpp:31-44: leaf holds 1 or 2 triangles
        BVHNode node;
        node.is_leaf = true;
        node.left = c[start].index;
        node.right = -1;
        AABB box = tri_box(tris[node.left]);
        if (end > start) {
            node.right = c[start + 1].index;
            expand_tri(box, tris[node.right]);
        }
        node.box = box;
        bvh.push_back(node);
        return;
    }
    AABB box = empty_box();                      // bvh.cpp:45-48: box of the centroids
    for (int i = start; i <= end; i++) expand(box, c[i].position);
    int axis = 0;                                // bvh.cpp:49-59: longest axis, strict >
    float max_extent = box.max.x - box.min.x;
    float extent = box.max.y - box.min.y;
    if (extent > max_extent) { axis = 1; max_extent = extent; }
    extent = box.max.z - box.min.z;
    if (extent > max_extent) { axis = 2; }
    int mid = (start + end) / 2;                 // bvh.cpp:60-64
    std::nth_element(c.begin() + start, c.begin() + mid, c.begin() + end + 1,
                     [axis](const Centroid& a, const Centroid& b) {
                         return get(a.position, axis) < get(b.position, axis);
                     });
    BVHNode node;                                // bvh.cpp:65-74: post-order, root last
    node.is_leaf = false;
    build_recursive(c, start, mid, bvh, tris);
    node.left = (int)bvh.size() - 1;
    build_recursive(c, mid + 1, end, bvh, tris);
    node.right = (int)bvh.size() - 1;
    expand(box, bvh[node.left].box);
    expand(box, bvh[node.right].box);
    node.box = box;
    bvh.push_back(node);
}

static int build_bvh(const Triangle* tris, int n, std::vector<BVHNode>& bvh) {   // bvh.cpp:77-98
    bvh.clear();
    std::vector<Centroid> c(n);
    for (int i = 0; i < n; i++) {
        // bvh.cpp:81-83: (v0 + v1 + v2) / 3.0f, where `/` multiplies by 1.f/3.0f
        c[i].position = divf(add(add(tris[i].v[0].position, tris[i].v[1].position), tris[i].v[2].position), 3.0f);
        c[i].index = i;
    }
    build_recursive(c, 0, n - 1, bvh, tris);
    return (int)bvh.size();
}

// ----------------------------------------------------------------------------
// Vertex transform -- cuda/transforms.cu:7-53
// ----------------------------------------------------------------------------
static void apply_transform(const f3& scale, const f3& translate, const f3& rot_deg, Triangle* tris, int n) {
    // transforms.cu:47: float * double(M_PI) / float, evaluated in double, stored as float
    f3 r = mk((float)(rot_deg.x * M_PI / 180.0f), (float)(rot_deg.y * M_PI / 180.0f), (float)(rot_deg.z * M_PI / 180.0f));
    float cosx = cosf(r.x), sinx = sinf(r.x);    // transforms.cu:8-19
    float cosy = cosf(r.y), siny = sinf(r.y);
    float cosz = cosf(r.z), sinz = sinf(r.z);
    float rot00 = cosy * cosz;
    float rot01 = -cosy * sinz;
    float rot02 = siny;
    float rot10 = sinx * siny * cosz + cosx * sinz;
    float rot11 = -sinx * siny * sinz + cosx * cosz;
    float rot12 = -sinx * cosy;
    float rot20 = -cosx * siny * cosz + sinx * sinz;
    float rot21 = cosx * siny * sinz + sinx * cosz;
    float rot22 = cosx * cosy;
    auto mmv = [&](const f3& v) {                // transforms.cu:24-30
        return mk(rot00 * v.x + rot01 * v.y + rot02 * v.z,
                  rot10 * v.x + rot11 * v.y + rot12 * v.z,
                  rot20 * v.x + rot21 * v.y + rot22 * v.z);
    };
    for (int t = 0; t < n; t++) {                // transforms.cu:32-43
        for (int i = 0; i < 3; i++) {
            Vertex& v = tris[t].v[i];
            v.position = mul3(v.position, scale);
            v.position = mmv(v.position);
            v.position = add(v.position, translate);
            v.normal = mmv(v.normal);
        }
    }
}

// ----------------------------------------------------------------------------
// Counter-based RNG (replaces cuRAND XORWOW by specification)
// ----------------------------------------------------------------------------
static inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Draw addressing.  Every random number a path uses has a fixed address, so a path can be evaluated by
// any lane of any launch in any order:
//   counter = (pixel, index, slot, stream), key = (seed low word, seed high word)
//   stream 1, index = semantic batch, slot 0: words 0,1 = the primary-ray jitter (u, v) of that batch
//   stream 0, index = global sample, slot = bounce * 65536 + k, bounce = scatters done so far on the path:
//       k = 0      : word 0 = Russian-roulette draw (raytrace.cu:230); word 1 = the material's own
//                    decision draw (metal lobe choice :149, refractive reflect-or-refract :186)
//       k = 1 + a  : words 0,1,2 = candidate number a of the unit-sphere rejection loop (:7-13)
// Each draw maps uint32 -> (0, 1] with cuRAND's curand_uniform formula.
static inline float u01(uint32_t x) { return (float)x * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }

struct Rng {
    uint32_t key[2];
    uint32_t pixel, index;
    void init(uint64_t seed, uint32_t pixel_, uint32_t index_) {
        key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
        pixel = pixel_; index = index_;
    }
    void block(uint32_t slot, uint32_t stream, uint32_t out[4]) const {
        const uint32_t ctr[4] = {pixel, index, slot, stream};
        philox4x32_10(ctr, key, out);
    }
};

// ----------------------------------------------------------------------------
// Camera -- cuda/camera.cu:23-64
// ----------------------------------------------------------------------------
struct Camera { f3 pos, lower_left, dx, dy; };

static Camera camera_setup(int width, int height, f3 pos, f3 dir, f3 up, float fov) {   // camera.cu:49-59
    dir = normalize(dir);
    up = normalize(up);
    f3 right = cross(dir, up);
    up = cross(right, dir);
    float aspect_ratio = (float)width / (float)height;
    float half_height = tanf(fov / 2.0f);
    float half_width = aspect_ratio * half_height;
    Camera c;
    c.pos = pos;
    c.lower_left = sub(sub(dir, mulf(right, half_width)), mulf(up, half_height));
    c.dx = mulf(right, 2.0f * half_width);
    c.dy = mulf(up, 2.0f * half_height);
    return c;
}

static inline Ray primary_ray(const Camera& c, int width, int height, int x, int y, float ju, float jv) {   // camera.cu:35-40
    float u = ((float)x + ju) / (float)width;
    float v = ((float)y + jv) / (float)height;
    f3 d = normalize(add(add(c.lower_left, mulf(c.dx, u)), mulf(c.dy, v)));
    return Ray{c.pos, d};
}

// ----------------------------------------------------------------------------
// Intersection -- include/bvh.h:39-58, cuda/raytrace.cu:15-78
// ----------------------------------------------------------------------------
struct HitRecord {   // raytrace.cu:15-22
    f3 point, normal, color;
    float t, u, v;
    const Triangle* triangle;
    int material_id;
    int object_index, triangle_index;   // bookkeeping for the parity tests only
};

struct Counters { uint64_t rays, box_tests, tri_tests, shaded, tex_fetches; };

static inline bool aabb_intersect(const AABB& b, const Ray& r, float& t_near, float& t_far) {   // bvh.h:39-58
    float t0 = -INFINITY, t1 = INFINITY;
    for (int i = 0; i < 3; i++) {
        float inv_d = 1.0f / get(r.direction, i);
        float t0_ = (get(b.min, i) - get(r.origin, i)) * inv_d;
        float t1_ = (get(b.max, i) - get(r.origin, i)) * inv_d;
        if (inv_d < 0.0f) { float tmp = t0_; t0_ = t1_; t1_ = tmp; }
        t0 = fmaxf(t0, t0_);
        t1 = fminf(t1, t1_);
        if (t0 > t1) return false;
    }
    t_near = t0;
    t_far = t1;
    return t_far > 0.0f;
}

static inline bool triangle_intersect(const Triangle& tri, const Ray& r, HitRecord& rec) {   // raytrace.cu:24-42
    f3 e1 = sub(tri.v[1].position, tri.v[0].position);
    f3 e2 = sub(tri.v[2].position, tri.v[0].position);
    f3 p = cross(r.direction, e2);
    float det = dot3(e1, p);
    if (det > -1e-6f && det < 1e-6f) return false;
    float inv_det = 1.0f / det;
    f3 tvec = sub(r.origin, tri.v[0].position);
    float u = dot3(tvec, p) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    f3 q = cross(tvec, e1);
    float v = dot3(r.direction, q) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot3(e2, q) * inv_det;
    if (t < 0.0f || t > rec.t) return false;
    rec.t = t; rec.u = u; rec.v = v;
    rec.point = add(r.origin, mulf(r.direction, t));   // primitives.cuh:17-19
    return true;
}

static const int MAX_HIT_STACK_SIZE = 32;   // common.cuh:17

static inline bool object_intersect(const Object& obj, int obj_index, const Ray& r, HitRecord& rec, Counters* cnt) {   // raytrace.cu:44-78
    int stack[MAX_HIT_STACK_SIZE];
    int top = 1;
    int hit_triangle = -1;
    stack[0] = obj.n_bvh_nodes - 1;
    while (top > 0) {
        int node_idx = stack[--top];
        const BVHNode& node = obj.bvh[node_idx];
        float t_near, t_far;
        if (cnt) cnt->box_tests++;
        if (!aabb_intersect(node.box, r, t_near, t_far)) continue;
        if (t_near > rec.t) continue;
        if (node.is_leaf) {
            if (node.left >= 0) {
                if (cnt) cnt->tri_tests++;
                if (triangle_intersect(obj.triangles[node.left], r, rec)) hit_triangle = node.left;
            }
            if (node.right >= 0) {
                if (cnt) cnt->tri_tests++;
                if (triangle_intersect(obj.triangles[node.right], r, rec)) hit_triangle = node.right;
            }
        } else {
            if (node.left >= 0 && top < MAX_HIT_STACK_SIZE) stack[top++] = node.left;
            if (node.right >= 0 && top < MAX_HIT_STACK_SIZE) stack[top++] = node.right;
        }
    }
    if (hit_triangle >= 0) {
        rec.triangle = &obj.triangles[hit_triangle];
        rec.material_id = rec.triangle->material_id;
        rec.object_index = obj_index;
        rec.triangle_index = hit_triangle;
        return true;
    }
    return false;
}

static inline bool scene_intersect(const Scene& s, const Ray& r, HitRecord& rec, Counters* cnt) {   // raytrace.cu:214-219
    rec.t = INFINITY;
    bool hit = false;
    if (cnt) cnt->rays++;
    for (size_t j = 0; j < s.objects.size(); j++) hit |= object_intersect(s.objects[j], (int)j, r, rec, cnt);
    return hit;
}

// ----------------------------------------------------------------------------
// Texture fetch (replaces tex2D<float4> by specification) -- sceneloader.cpp:240-246
// normalized coordinates, linear filter, border addressing, normalized-float read.
// ----------------------------------------------------------------------------
static inline void texel(const Texture& t, int i, int j, float out[3]) {
    if (i < 0 || j < 0 || i >= t.width || j >= t.height) { out[0] = out[1] = out[2] = 0.0f; return; }
    const uint8_t* p = t.rgba + 4 * ((size_t)j * t.width + i);
    out[0] = (float)p[0] / 255.0f; out[1] = (float)p[1] / 255.0f; out[2] = (float)p[2] / 255.0f;
}

static inline f3 tex2d(const Texture& t, float u, float v) {
    float xb = u * (float)t.width - 0.5f;
    float yb = v * (float)t.height - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    // 8 fractional bits, round to nearest
    float a = floorf((xb - fi) * 256.0f + 0.5f) / 256.0f;
    float b = floorf((yb - fj) * 256.0f + 0.5f) / 256.0f;
    // clamp far-out coordinates before the int cast; anything outside is border anyway
    fi = fminf(fmaxf(fi, -2.0f), (float)t.width + 1.0f);
    fj = fminf(fmaxf(fj, -2.0f), (float)t.height + 1.0f);
    if (!(fi == fi)) fi = -2.0f;
    if (!(fj == fj)) fj = -2.0f;
    int i = (int)fi, j = (int)fj;
    float t00[3], t10[3], t01[3], t11[3];
    texel(t, i, j, t00); texel(t, i + 1, j, t10); texel(t, i, j + 1, t01); texel(t, i + 1, j + 1, t11);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    float o[3];
    for (int c = 0; c < 3; c++) o[c] = w00 * t00[c] + w10 * t10[c] + w01 * t01[c] + w11 * t11[c];
    return mk(o[0], o[1], o[2]);
}

// ----------------------------------------------------------------------------
// Shading -- cuda/raytrace.cu:7-13, 80-195
// ----------------------------------------------------------------------------
static inline f3 random_in_unit_sphere3(const Rng& rng, uint32_t bounce) {   // raytrace.cu:7-13
    f3 p;
    uint32_t attempt = 0;
    do {
        uint32_t w[4];
        rng.block(bounce * 65536u + 1u + attempt, 0, w);
        p = sub(mulf(mk(u01(w[0]), u01(w[1]), u01(w[2])), 2.0f), mk(1.0f, 1.0f, 1.0f));
        attempt++;
    } while (dot3(p, p) >= 1.0f && attempt < 65535u);
    return p;
}

static inline void triangle_interpolate(const Scene& s, HitRecord& rec, Counters* cnt) {   // raytrace.cu:80-96
    float u = rec.u, v = rec.v;
    float w = 1.0f - u - v;
    const Triangle& tri = *rec.triangle;
    f3 norm = add(add(mulf(tri.v[0].normal, w), mulf(tri.v[1].normal, u)), mulf(tri.v[2].normal, v));
    rec.normal = normalize(norm);
    const Material& m = s.materials[rec.material_id];
    if (m.texture < 0) {
        rec.color = m.albedo;
    } else {
        float tx = w * tri.v[0].texcoord.x + u * tri.v[1].texcoord.x + v * tri.v[2].texcoord.x;
        float ty = w * tri.v[0].texcoord.y + u * tri.v[1].texcoord.y + v * tri.v[2].texcoord.y;
        if (cnt) cnt->tex_fetches++;
        rec.color = tex2d(s.textures[m.texture], tx, ty);
    }
}

static inline f3 reflect(const f3& v, const f3& n) {   // raytrace.cu:98-100
    return sub(v, mulf(n, 2.0f * dot3(v, n)));
}

static inline bool refract(const f3& v, const f3& n, float ni_over_nt, f3& refracted) {   // raytrace.cu:102-110
    float dt = dot3(v, n);
    float discriminant = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (discriminant > 0.0f) {
        refracted = sub(mulf(sub(v, mulf(n, dt)), ni_over_nt), mulf(n, sqrtf(discriminant)));
        return true;
    }
    return false;
}

// CUDA resolves `pow(float, int)` in device code to its integer-power routine
// (repeated squaring in binary32), not to powf; for exponent 5 that is
// a * ((a*a) * (a*a)).
static inline float powi5(float a) {
    float r = a;          // bit 0 of 5
    a = a * a;            // a^2
    a = a * a;            // a^4 (bit 2 of 5)
    return r * a;
}

static inline float schlick(float cosine, float ref_idx) {   // raytrace.cu:112-116
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powi5(1.0f - cosine);
}

static inline Ray offset_ray(const HitRecord& rec, const f3& dir) {   // raytrace.cu:118
    return Ray{add(rec.point, mulf(dir, 1e-4f)), dir};
}

// `decision` is the bounce's material decision draw (word 1 of the bounce's slot-0 block).
static inline bool material_scatter(const Rng& rng, uint32_t bounce, float decision, const Material& m, const Ray& r,
                                    const HitRecord& rec, Ray& wo, f3& color) {   // raytrace.cu:120-195
    switch (m.type) {
    case LAMBERTIAN: {
        f3 direction = rec.normal;
        if (dot3(r.direction, rec.normal) > 0.0f) direction = neg(rec.normal);
        direction = normalize(add(direction, random_in_unit_sphere3(rng, bounce)));
        float cos_theta = fabsf(dot3(direction, rec.normal));
        color = mul3(color, mulf(rec.color, cos_theta));
        wo = offset_ray(rec, direction);
        return true;
    }
    case METAL: {
        f3 normal = rec.normal;
        if (dot3(r.direction, rec.normal) > 0.0f) normal = neg(rec.normal);
        f3 random = random_in_unit_sphere3(rng, bounce);
        f3 direction;
        if (4.0f * decision < 3.0f + m.metal_fuzz) {
            direction = add(normal, random);
            float cos_theta = fabsf(dot3(direction, normal));
            color = mul3(color, mulf(rec.color, cos_theta));
        } else {
            direction = reflect(r.direction, normal);
            direction = add(normalize(direction), mulf(random, m.metal_fuzz));
        }
        wo = offset_ray(rec, direction);
        return true;
    }
    case REFLECTIVE: {
        f3 direction = reflect(r.direction, rec.normal);
        color = mul3(color, rec.color);
        wo = offset_ray(rec, direction);
        return true;
    }
    case REFRACTIVE: {
        f3 outward_normal;
        float ni_over_nt, cosine;
        // The reference leaves `direction` uninitialised when refraction fails and the
        // uniform draw equals exactly 1.0 (probability ~2^-25 per such event); it is defined here as
        // the reflection, the value every other draw gives in that case.
        f3 direction = mk(0.0f, 0.0f, 0.0f);
        if (dot3(r.direction, rec.normal) > 0.0f) {
            outward_normal = neg(rec.normal);
            ni_over_nt = m.optical_density;
            cosine = m.optical_density * dot3(r.direction, rec.normal);
        } else {
            outward_normal = rec.normal;
            ni_over_nt = 1.0f / m.optical_density;
            cosine = -dot3(r.direction, rec.normal);
        }
        float reflect_prob;
        const bool refracted = refract(r.direction, outward_normal, ni_over_nt, direction);
        if (refracted) reflect_prob = schlick(cosine, m.optical_density);
        else reflect_prob = 1.0f;
        if (decision < reflect_prob || !refracted) direction = reflect(r.direction, rec.normal);
        color = mul3(color, rec.color);
        wo = offset_ray(rec, direction);
        return true;
    }
    default:
        return false;
    }
}

// One camera sample -- the body of the spp loop, raytrace.cu:210-244.
// Returns false when the path ran out of depth (contributes nothing, :243).
static inline bool trace_path(const Scene& s, Ray ray, const Rng& rng, int max_depth, const f3& ambient,
                              float russian_roulette, f3& color_out, Counters* cnt) {
    f3 color = mk(1.0f, 1.0f, 1.0f);
    int depth = max_depth;
    uint32_t bounce = 0;
    while (depth--) {
        HitRecord rec;
        if (!scene_intersect(s, ray, rec, cnt)) { color = mul3(color, ambient); break; }
        if (cnt) cnt->shaded++;
        triangle_interpolate(s, rec, cnt);
        const Material& m = s.materials[rec.material_id];
        if (m.type == LIGHT) { color = mul3(color, rec.color); break; }
        uint32_t w[4];
        rng.block(bounce * 65536u, 0, w);
        if (u01(w[0]) < russian_roulette) {
            Ray new_ray;
            if (!material_scatter(rng, bounce, u01(w[1]), m, ray, rec, new_ray, color)) { color = mul3(color, ambient); break; }
            ray = new_ray;
            bounce++;
        } else { color = mul3(color, ambient); break; }
    }
    if (depth < 0) return false;
    color_out = color;
    return true;
}

}  // namespace orc

// ============================================================================
// C entry points (ctypes-friendly).  All pointers are host memory.
// ============================================================================
using namespace orc;

extern "C" {

struct orc_render_params {
    int32_t width, height;
    int32_t max_depth;
    int32_t semantic_batch;      // samples that share one jittered primary ray (worker.cpp:71-91 batch loop)
    float fov;                   // radians
    float russian_roulette;
    float background[3];
    float camera_pos[3], camera_dir[3], camera_up[3];
    uint64_t seed;
};

int orc_sizeof(int what) {
    switch (what) {
    case 0: return (int)sizeof(Triangle);
    case 1: return (int)sizeof(BVHNode);
    case 2: return (int)sizeof(Material);
    case 3: return (int)sizeof(Ray);
    case 4: return (int)sizeof(orc_render_params);
    default: return -1;
    }
}

int orc_build_bvh(const void* tris, int n, void* out_nodes, int cap) {
    std::vector<BVHNode> bvh;
    int m = build_bvh((const Triangle*)tris, n, bvh);
    if (m > cap) return -m;
    memcpy(out_nodes, bvh.data(), (size_t)m * sizeof(BVHNode));
    return m;
}

void orc_apply_transform(const float* scale, const float* translate, const float* rot_deg, void* tris, int n) {
    apply_transform(mk(scale[0], scale[1], scale[2]), mk(translate[0], translate[1], translate[2]),
                    mk(rot_deg[0], rot_deg[1], rot_deg[2]), (Triangle*)tris, n);
}

void orc_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }

// The four (0,1] draws of one addressed block.
void orc_uniforms(uint64_t seed, uint32_t pixel, uint32_t index, uint32_t slot, uint32_t stream, float* out4) {
    Rng r; r.init(seed, pixel, index);
    uint32_t w[4];
    r.block(slot, stream, w);
    for (int i = 0; i < 4; i++) out4[i] = u01(w[i]);
}

void* orc_scene_create(int n_objects, void** tris, const int* n_tris, void** nodes, const int* n_nodes,
                       int n_materials, const void* materials, int n_textures, const uint8_t** tex_rgba,
                       const int* tex_w, const int* tex_h) {
    Scene* s = new Scene();
    for (int i = 0; i < n_objects; i++) s->objects.push_back(Object{(Triangle*)tris[i], (BVHNode*)nodes[i], n_tris[i], n_nodes[i]});
    s->materials.assign((const Material*)materials, (const Material*)materials + n_materials);
    for (int i = 0; i < n_textures; i++) s->textures.push_back(Texture{tex_rgba[i], tex_w[i], tex_h[i]});
    return s;
}

void orc_scene_destroy(void* scene) { delete (Scene*)scene; }

// camera.cu:49-59 -> out[12] = pos, lower_left, dx, dy
void orc_camera_setup(int w, int h, const float* pos, const float* dir, const float* up, float fov, float* out) {
    Camera c = camera_setup(w, h, mk(pos[0], pos[1], pos[2]), mk(dir[0], dir[1], dir[2]), mk(up[0], up[1], up[2]), fov);
    memcpy(out, &c, sizeof(c));
}

// The jittered primary rays of one semantic batch (camera.cu:23-41): rays[6*idx], idx = y*w + x.
void orc_primary_rays(const orc_render_params* p, uint32_t batch_index, float* rays) {
    Camera c = camera_setup(p->width, p->height, mk(p->camera_pos[0], p->camera_pos[1], p->camera_pos[2]),
                            mk(p->camera_dir[0], p->camera_dir[1], p->camera_dir[2]),
                            mk(p->camera_up[0], p->camera_up[1], p->camera_up[2]), p->fov);
    for (int y = 0; y < p->height; y++)
        for (int x = 0; x < p->width; x++) {
            int idx = y * p->width + x;
            Rng j; j.init(p->seed, (uint32_t)idx, batch_index);
            uint32_t w[4];
            j.block(0, 1, w);
            Ray r = primary_ray(c, p->width, p->height, x, y, u01(w[0]), u01(w[1]));
            memcpy(rays + 6 * (size_t)idx, &r, sizeof(r));
        }
}

// Closest hit for explicit rays: object index, triangle index within the object, t, u, v (-1 / inf on miss).
void orc_closest_hit(const void* scene, int n_rays, const float* rays, int32_t* obj, int32_t* tri, float* t, float* u, float* v,
                     uint64_t* counters /* [5] or null */) {
    const Scene& s = *(const Scene*)scene;
    Counters total{0, 0, 0, 0, 0};
#pragma omp parallel
    {
        Counters c{0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 256)
        for (int i = 0; i < n_rays; i++) {
            Ray r; memcpy(&r, rays + 6 * (size_t)i, sizeof(r));
            HitRecord rec;
            rec.u = rec.v = 0.0f;
            bool hit = scene_intersect(s, r, rec, &c);
            obj[i] = hit ? rec.object_index : -1;
            tri[i] = hit ? rec.triangle_index : -1;
            t[i] = rec.t;
            if (u) u[i] = hit ? rec.u : 0.0f;
            if (v) v[i] = hit ? rec.v : 0.0f;
        }
#pragma omp critical
        { total.rays += c.rays; total.box_tests += c.box_tests; total.tri_tests += c.tri_tests; }
    }
    if (counters) { counters[0] = total.rays; counters[1] = total.box_tests; counters[2] = total.tri_tests; counters[3] = 0; counters[4] = 0; }
}

// Exhaustive nearest hit (no BVH): every triangle of every object through triangle_intersect.
// A consistency check for the traversal above; its t equals the traversal's t whenever no
// slab test rejects a box whose triangle Moller-Trumbore accepts.
void orc_closest_hit_brute(const void* scene, int n_rays, const float* rays, float* t) {
    const Scene& s = *(const Scene*)scene;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n_rays; i++) {
        Ray r; memcpy(&r, rays + 6 * (size_t)i, sizeof(r));
        HitRecord rec;
        rec.t = INFINITY;
        for (const Object& o : s.objects)
            for (int k = 0; k < o.n_triangles; k++) triangle_intersect(o.triangles[k], r, rec);
        t[i] = rec.t;
    }
}

// The Worker batch loop (worker.cpp:69-91) + raytrace_kernel (raytrace.cu:197-247) for global samples
// [first_sample, first_sample + n_samples): per pixel and per semantic batch, one jittered primary ray,
// the batch's samples summed in order into a batch total, the batch total added to accum[3*idx].
// first_sample must be a multiple of semantic_batch.  counters: rays, box tests, triangle tests, shaded hits, texture fetches.
void orc_render(const void* scene, const orc_render_params* p, int first_sample, int n_samples, float* accum, uint64_t* counters) {
    const Scene& s = *(const Scene*)scene;
    Camera c = camera_setup(p->width, p->height, mk(p->camera_pos[0], p->camera_pos[1], p->camera_pos[2]),
                            mk(p->camera_dir[0], p->camera_dir[1], p->camera_dir[2]),
                            mk(p->camera_up[0], p->camera_up[1], p->camera_up[2]), p->fov);
    f3 ambient = mk(p->background[0], p->background[1], p->background[2]);
    const int B = p->semantic_batch;
    Counters total{0, 0, 0, 0, 0};
#pragma omp parallel
    {
        Counters cnt{0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 1)
        for (int y = 0; y < p->height; y++) {
            for (int x = 0; x < p->width; x++) {
                int idx = y * p->width + x;
                f3 out = mk(accum[3 * (size_t)idx], accum[3 * (size_t)idx + 1], accum[3 * (size_t)idx + 2]);
                int sample = first_sample, end = first_sample + n_samples;
                while (sample < end) {
                    uint32_t batch = (uint32_t)(sample / B);
                    int batch_end = std::min(end, (int)(batch + 1) * B);
                    Rng j; j.init(p->seed, (uint32_t)idx, batch);
                    uint32_t jw[4];
                    j.block(0, 1, jw);
                    Ray ray = primary_ray(c, p->width, p->height, x, y, u01(jw[0]), u01(jw[1]));
                    f3 total_color = mk(0.0f, 0.0f, 0.0f);   // raytrace.cu:208
                    for (; sample < batch_end; sample++) {
                        Rng rng; rng.init(p->seed, (uint32_t)idx, (uint32_t)sample);
                        f3 color;
                        if (trace_path(s, ray, rng, p->max_depth, ambient, p->russian_roulette, color, &cnt))
                            total_color = add(total_color, color);   // raytrace.cu:244
                    }
                    out = add(out, total_color);              // raytrace.cu:246
                }
                accum[3 * (size_t)idx] = out.x; accum[3 * (size_t)idx + 1] = out.y; accum[3 * (size_t)idx + 2] = out.z;
            }
        }
#pragma omp critical
        {
            total.rays += cnt.rays; total.box_tests += cnt.box_tests; total.tri_tests += cnt.tri_tests;
            total.shaded += cnt.shaded; total.tex_fetches += cnt.tex_fetches;
        }
    }
    if (counters) {
        counters[0] = total.rays; counters[1] = total.box_tests; counters[2] = total.tri_tests;
        counters[3] = total.shaded; counters[4] = total.tex_fetches;
    }
}

// raytrace.cu:249-261
void orc_post_process(int n_pixels, int spp, float gamma, float* buf) {
    for (int i = 0; i < n_pixels; i++) {
        f3 c = mk(buf[3 * (size_t)i], buf[3 * (size_t)i + 1], buf[3 * (size_t)i + 2]);
        c = divf(c, (float)spp);
        c.x = fminf(1.0f, fmaxf(0.0f, c.x));
        c.y = fminf(1.0f, fmaxf(0.0f, c.y));
        c.z = fminf(1.0f, fmaxf(0.0f, c.z));
        c.x = powf(c.x, 1.0f / gamma);
        c.y = powf(c.y, 1.0f / gamma);
        c.z = powf(c.z, 1.0f / gamma);
        buf[3 * (size_t)i] = c.x; buf[3 * (size_t)i + 1] = c.y; buf[3 * (size_t)i + 2] = c.z;
    }
}

void orc_tex2d(const uint8_t* rgba, int w, int h, int n, const float* uv, float* out) {
    Texture t{rgba, w, h};
    for (int i = 0; i < n; i++) {
        f3 c = tex2d(t, uv[2 * i], uv[2 * i + 1]);
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

// Scatter known-answer helper: one material_scatter call with an explicit uniform stream position.
// in: ray (6), hit point (3), normal (3), color (3), carried color (3); out: new ray (6), color (3); returns 0/1.
int orc_scatter(const void* material, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, const float* ray, const float* point,
                const float* normal, const float* hit_color, const float* color_in, float* ray_out, float* color_out) {
    Rng rng; rng.init(seed, pixel, sample);
    uint32_t w[4];
    rng.block(bounce * 65536u, 0, w);
    Ray r; memcpy(&r, ray, sizeof(r));
    HitRecord rec;
    rec.point = mk(point[0], point[1], point[2]);
    rec.normal = mk(normal[0], normal[1], normal[2]);
    rec.color = mk(hit_color[0], hit_color[1], hit_color[2]);
    f3 color = mk(color_in[0], color_in[1], color_in[2]);
    Ray wo{mk(0, 0, 0), mk(0, 0, 0)};
    bool ok = material_scatter(rng, bounce, u01(w[1]), *(const Material*)material, r, rec, wo, color);
    memcpy(ray_out, &wo, sizeof(wo));
    color_out[0] = color.x; color_out[1] = color.y; color_out[2] = color.z;
    return ok ? 1 : 0;
}

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
