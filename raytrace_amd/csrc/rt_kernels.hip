// rt_kernels.hip -- gfx950 kernels of the path-tracing core and the C ABI around them.
//
// Kernel design (wave64, 256 CUs, no MFMA: nothing here is a dense contraction):
//  * one lane owns one pixel and walks that pixel's samples in order, as a flat
//    state machine: every trip of the main loop traces ONE ray segment for every
//    live lane, whatever sample/bounce that lane is on, so a lane whose path
//    ended starts its next sample at once instead of idling until the slowest
//    path of the wave ends (upstream's nested spp/depth loops,
//    cuda/raytrace.cu:209-245, reconverge at every sample);
//  * a wave covers an 8x8 pixel tile, a 256-thread workgroup a 16x16 tile, and
//    workgroups are dealt so that each XCD's L2 serves one band of the image;
//  * traversal visits nodes in upstream's order (objects ascending, right child
//    first, leaf left triangle then right, prune on t_near > best t), which
//    makes hit records bit-identical, but it reads one 64-byte record per inner
//    node that holds both children's boxes, hoists 1/dir out of the slab test,
//    keeps its stack in LDS (one 8-byte column per lane, conflict-free) and
//    lets every lane advance through the object list on its own;
//  * random numbers are Philox4x32-10 keyed by (seed; pixel, sample, block,
//    stream): no state array, results independent of launch batching.
//
// Built with -ffp-contract=off; expressions keep upstream's operand order.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_host.h"

using namespace rtc;

// ---------------------------------------------------------------------------
// device-side scene view and parameters
// ---------------------------------------------------------------------------
struct SceneDev {
    ChildRec root;            // box + reference of the whole scene (root of the object-level tree)
    const Node64* nodes;
    const TriIsect* isect;
    const TriShade* shade;
    const TriOrigin* origin;
    const MatDev* mats;
    const TexDev* tex;
    const uint32_t* texels;
    int n_materials;
};

struct PathParams {
    CameraFrame cam;
    int n_pixel;
    int first_sample, n_samples, batch;   // this pass: global samples [first_sample, first_sample + n_samples)
    int chunk, n_chunks;                  // samples per lane; lanes (chunks) per pixel in this pass
    int max_depth;
    float ambient[3];
    float rr;
    uint32_t key0, key1;
    float* slots;                         // [n_samples][n_pixel][3]: radiance of each path of the pass
    unsigned long long* counters;
    int tiles_x, tiles_y;
    int xcd_remap;
    int lds_slots;                        // stack levels kept in LDS
};

#define RT_BLOCK 256
#define RT_INF __builtin_inff()

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 vadd(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return mk3(a.x + (-b.x), a.y + (-b.y), a.z + (-b.z)); }
__device__ __forceinline__ v3 vscale(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 vmul(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 vneg(v3 a) { return vscale(a, -1.0f); }                       // common.cuh:126-130
__device__ __forceinline__ float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 vcross(v3 a, v3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ v3 vnormalize(v3 a) {                                              // common.cuh:75-78
    float n = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return mk3(a.x / n, a.y / n, a.z / n);
}

// ---------------------------------------------------------------------------
// Philox4x32-10 with addressed draws: counter = (pixel, index, slot, stream), key = seed.
//   stream 1, index = semantic batch, slot 0: words 0,1 = primary-ray jitter
//   stream 0, index = global sample, slot = bounce * 65536 + k:
//       k = 0: word 0 = Russian roulette, word 1 = the material's decision draw
//       k = 1 + a: words 0..2 = candidate a of the unit-sphere rejection loop
// No generator state lives in registers, and every lane of a wave calls the generator at the same
// program points.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t k0, uint32_t k1,
                                       uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, x0), lo0 = 0xD2511F53u * x0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, x2), lo1 = 0xCD9E8D57u * x2;
        const uint32_t n0 = hi1 ^ x1 ^ k0;
        const uint32_t n2 = hi0 ^ x3 ^ k1;
        x0 = n0; x1 = lo1; x2 = n2; x3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = x0; o1 = x1; o2 = x2; o3 = x3;
}

// cuRAND's uint32 -> (0,1] mapping
__device__ __forceinline__ float u01(uint32_t v) { return (float)v * 2.3283064365386963e-10f + 1.1641532182693481e-10f; }

// ---------------------------------------------------------------------------
// traversal
// ---------------------------------------------------------------------------
struct RayD {
    float ox, oy, oz;
    float dx, dy, dz;
    float ix, iy, iz;    // 1/d, hoisted out of include/bvh.h:43
};

__device__ __forceinline__ void ray_set(RayD& r, v3 o, v3 d) {
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    r.dx = d.x; r.dy = d.y; r.dz = d.z;
    r.ix = 1.0f / d.x; r.iy = 1.0f / d.y; r.iz = 1.0f / d.z;
}

// include/bvh.h:39-58.  The per-axis early exit there is equivalent to one final
// t0 > t1 test because t0 never decreases and t1 never increases.
__device__ __forceinline__ bool slab(float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                     const RayD& r, float& t_near) {
    const float ax = (mnx - r.ox) * r.ix, bx = (mxx - r.ox) * r.ix;
    const float ay = (mny - r.oy) * r.iy, by = (mxy - r.oy) * r.iy;
    const float az = (mnz - r.oz) * r.iz, bz = (mxz - r.oz) * r.iz;
    const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;
    float t0 = fmaxf(-RT_INF, nx ? bx : ax);
    float t1 = fminf(RT_INF, nx ? ax : bx);
    t0 = fmaxf(t0, ny ? by : ay);
    t1 = fminf(t1, ny ? ay : by);
    t0 = fmaxf(t0, nz ? bz : az);
    t1 = fminf(t1, nz ? az : bz);
    t_near = t0;
    return !(t0 > t1) && (t1 > 0.0f);
}

struct Hit {
    float t, u, v;
    uint32_t tri;      // packed triangle index, REF_NONE when nothing was hit
};

// cuda/raytrace.cu:24-42 on a packed record (e1, e2 precomputed).  Rejections keep
// upstream's comparison polarity so that NaNs fall the same way.
__device__ __forceinline__ void tri_test(const TriIsect* __restrict__ tris, uint32_t p, const RayD& r, Hit& h) {
    const float4* q = reinterpret_cast<const float4*>(tris + p);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2];
    const float v0x = q0.x, v0y = q0.y, v0z = q0.z;
    const float e1x = q0.w, e1y = q1.x, e1z = q1.y;
    const float e2x = q1.z, e2y = q1.w, e2z = q2.x;
    const float px = r.dy * e2z - r.dz * e2y;
    const float py = r.dz * e2x - r.dx * e2z;
    const float pz = r.dx * e2y - r.dy * e2x;
    const float det = e1x * px + e1y * py + e1z * pz;
    bool rej = (det > -1e-6f) && (det < 1e-6f);
    const float inv_det = 1.0f / det;
    const float tx = r.ox + (-v0x), ty = r.oy + (-v0y), tz = r.oz + (-v0z);
    const float u = (tx * px + ty * py + tz * pz) * inv_det;
    rej = rej || (u < 0.0f) || (u > 1.0f);
    const float qx = ty * e1z - tz * e1y;
    const float qy = tz * e1x - tx * e1z;
    const float qz = tx * e1y - ty * e1x;
    const float v = (r.dx * qx + r.dy * qy + r.dz * qz) * inv_det;
    rej = rej || (v < 0.0f) || (u + v > 1.0f);
    const float t = (e2x * qx + e2y * qy + e2z * qz) * inv_det;
    rej = rej || (t < 0.0f) || (t > h.t);
    if (!rej) { h.t = t; h.u = u; h.v = v; h.tri = p; }
}

// Closest hit over the whole scene in upstream's visiting order (cuda/raytrace.cu:44-78, 214-219):
// right child first, leaf left triangle then right, prune on t_near > best t when a node comes off the
// stack.  The objects hang under the object-level tree, so there is one walk, not one per object.
// `stack` points at this lane's column of the LDS stack; consecutive slots are `stride` entries apart.
// The first `lds_slots` stack levels live in LDS; deeper levels (rare: a level is only used while BOTH
// children of a node are hit) spill to a per-lane private array, so LDS no longer caps occupancy.
#define RT_SPILL_SLOTS 56
typedef __attribute__((address_space(3))) uint2 lds_uint2;   // explicitly LDS: never merged with the private spill into a flat access
template <bool COUNT>
__device__ __forceinline__ void closest_hit(const SceneDev& sc, const RayD& r, Hit& h, lds_uint2* stack, int stride, int lds_slots,
                                            uint32_t& n_box, uint32_t& n_tri, uint32_t& max_sp) {
    h.t = RT_INF; h.u = 0.0f; h.v = 0.0f; h.tri = REF_NONE;
    uint2 spill[RT_SPILL_SLOTS];
    int sp = 0;
    uint32_t cur = REF_NONE;
    {
        float tn;
        if (COUNT) n_box++;
        if (slab(sc.root.bmin[0], sc.root.bmin[1], sc.root.bmin[2], sc.root.bmax[0], sc.root.bmax[1], sc.root.bmax[2], r, tn))
            cur = sc.root.ref;
    }
    // next deferred sibling whose t_near does not exceed the best t (upstream tests this at pop time, raytrace.cu:55)
    auto pop = [&]() -> uint32_t {
        while (sp > 0) {
            --sp;
            uint2 e;
            if (sp < lds_slots) { e.x = stack[sp * stride].x; e.y = stack[sp * stride].y; }
            else e = spill[sp - lds_slots];
            if (!(__uint_as_float(e.y) > h.t)) return e.x;
        }
        return REF_NONE;
    };
    for (;;) {
        // inner nodes: one 64-byte record decides both children.  REF_NONE has the leaf bit set and ends the loop.
        while (!(cur & REF_LEAF)) {
            const float4* q = reinterpret_cast<const float4*>(sc.nodes + cur);
            const float4 a = q[0], b = q[1], c = q[2], d = q[3];
            float tnl, tnr;
            if (COUNT) n_box += 2;
            const bool hl = slab(a.x, a.y, a.z, a.w, b.x, b.y, r, tnl) && !(tnl > h.t);
            const bool hr = slab(c.x, c.y, c.z, c.w, d.x, d.y, r, tnr) && !(tnr > h.t);
            const uint32_t refl = __float_as_uint(b.z), refr = __float_as_uint(d.z);
            // upstream pushes left then right, so the right child is visited first.  A left child whose
            // t_near already exceeds the best t can be dropped now: the best t only shrinks.
            if (hr) {
                if (hl) {
                    const uint2 e = make_uint2(refl, __float_as_uint(tnl));
                    if (sp < lds_slots) { stack[sp * stride].x = e.x; stack[sp * stride].y = e.y; }
                    else spill[sp - lds_slots] = e;
                    sp++;
                    if (COUNT) max_sp = max(max_sp, (uint32_t)sp);
                }
                cur = refr;
            } else if (hl) {
                cur = refl;
            } else {
                cur = pop();
            }
        }
        if (cur == REF_NONE) break;
        const uint32_t p = cur & REF_MASK;
        if (COUNT) n_tri += (cur & REF_PAIR) ? 2u : 1u;
        tri_test(sc.isect, p, r, h);
        if (cur & REF_PAIR) tri_test(sc.isect, p + 1, r, h);
        cur = pop();
    }
}

// ---------------------------------------------------------------------------
// shading (cuda/raytrace.cu:7-13, 80-195)
// ---------------------------------------------------------------------------
__device__ __forceinline__ v3 unit_sphere(uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t k0, uint32_t k1) {   // raytrace.cu:7-13
    v3 p;
    uint32_t attempt = 0;
    do {
        uint32_t w0, w1, w2, w3;
        philox(pixel, sample, bounce * 65536u + 1u + attempt, 0u, k0, k1, w0, w1, w2, w3);
        p = mk3(u01(w0) * 2.0f + (-1.0f), u01(w1) * 2.0f + (-1.0f), u01(w2) * 2.0f + (-1.0f));
        attempt++;
    } while (vdot(p, p) >= 1.0f && attempt < 65535u);
    return p;
}

__device__ __forceinline__ v3 reflect3(v3 v, v3 n) { return vsub(v, vscale(n, 2.0f * vdot(v, n))); }   // raytrace.cu:98-100

__device__ __forceinline__ void texel_rgb(const SceneDev& sc, const TexDev& t, int i, int j, float& r, float& g, float& b) {
    if (i < 0 || j < 0 || i >= t.width || j >= t.height) { r = g = b = 0.0f; return; }   // border addressing
    const uint32_t px = sc.texels[t.offset + (uint32_t)j * (uint32_t)t.width + (uint32_t)i];
    r = (float)(px & 0xFFu) / 255.0f;
    g = (float)((px >> 8) & 0xFFu) / 255.0f;
    b = (float)((px >> 16) & 0xFFu) / 255.0f;
}

// Software stand-in for tex2D<float4> with normalized coordinates, linear filter,
// border addressing and 8-bit fractional weights (csrc/sceneloader.cpp:240-246).
__device__ __forceinline__ v3 tex_fetch(const SceneDev& sc, int tex_index, float u, float v) {
    const TexDev t = sc.tex[tex_index];
    const float xb = u * (float)t.width - 0.5f;
    const float yb = v * (float)t.height - 0.5f;
    float fi = floorf(xb), fj = floorf(yb);
    const float a = floorf((xb - fi) * 256.0f + 0.5f) / 256.0f;
    const float b = floorf((yb - fj) * 256.0f + 0.5f) / 256.0f;
    fi = fminf(fmaxf(fi, -2.0f), (float)t.width + 1.0f);
    fj = fminf(fmaxf(fj, -2.0f), (float)t.height + 1.0f);
    if (!(fi == fi)) fi = -2.0f;
    if (!(fj == fj)) fj = -2.0f;
    const int i = (int)fi, j = (int)fj;
    float r00, g00, b00, r10, g10, b10, r01, g01, b01, r11, g11, b11;
    texel_rgb(sc, t, i, j, r00, g00, b00);
    texel_rgb(sc, t, i + 1, j, r10, g10, b10);
    texel_rgb(sc, t, i, j + 1, r01, g01, b01);
    texel_rgb(sc, t, i + 1, j + 1, r11, g11, b11);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return mk3(w00 * r00 + w10 * r10 + w01 * r01 + w11 * r11,
               w00 * g00 + w10 * g10 + w01 * g01 + w11 * g11,
               w00 * b00 + w10 * b10 + w01 * b01 + w11 * b11);
}

// pow(float, int) in CUDA device code is repeated squaring in binary32; exponent 5 -> a * ((a*a)*(a*a))
__device__ __forceinline__ float powi5(float a) { const float a2 = a * a; const float a4 = a2 * a2; return a * a4; }

// ---------------------------------------------------------------------------
// wave-aggregated counters
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_add(unsigned long long* dst, uint32_t v) {
    unsigned long long s = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(dst, s);
}

// ---------------------------------------------------------------------------
// camera
// ---------------------------------------------------------------------------
__device__ __forceinline__ v3 primary_dir(const CameraFrame& cam, int x, int y, float ju, float jv) {   // camera.cu:35-39
    const float u = ((float)x + ju) / (float)cam.width;
    const float v = ((float)y + jv) / (float)cam.height;
    const v3 ll = mk3(cam.lower_left[0], cam.lower_left[1], cam.lower_left[2]);
    const v3 dx = mk3(cam.dx[0], cam.dx[1], cam.dx[2]);
    const v3 dy = mk3(cam.dy[0], cam.dy[1], cam.dy[2]);
    return vnormalize(vadd(vadd(ll, vscale(dx, u)), vscale(dy, v)));
}

// ---------------------------------------------------------------------------
// the path kernel
// ---------------------------------------------------------------------------
// One lane traces `chunk` consecutive samples of one pixel as a flat state machine and writes each
// finished path's radiance into that sample's slot; accumulate_kernel then sums the slots per pixel
// in upstream's order.  Work unit = (16x16 pixel tile, chunk of samples): many short workgroups,
// so neither a costly pixel nor a costly tile can leave the chip idle at the end of a launch.
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void path_kernel(SceneDev sc, PathParams pp) {
    extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];
    lds_uint2* stack = (lds_uint2*)lds_stack + threadIdx.x;

    int id = blockIdx.x;
    if (pp.xcd_remap) {
        // workgroups b and b+8 share an XCD: hand each of the 8 residue classes a contiguous run of
        // work units, i.e. one band of the image (bijective for any grid size)
        const int nb = gridDim.x, q = nb >> 3, rem = nb & 7, xc = id & 7;
        id = (xc < rem ? xc * (q + 1) : rem * (q + 1) + (xc - rem) * q) + (id >> 3);
    }
    const int tile = id / pp.n_chunks, chunk_id = id - tile * pp.n_chunks;
    const int tx = tile % pp.tiles_x, ty = tile / pp.tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = tx * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = (x < pp.cam.width) && (y < pp.cam.height);
    const int pixel = y * pp.cam.width + x;
    const uint32_t k0 = pp.key0, k1 = pp.key1;
    const v3 ambient = mk3(pp.ambient[0], pp.ambient[1], pp.ambient[2]);
    const v3 cam_pos = mk3(pp.cam.pos[0], pp.cam.pos[1], pp.cam.pos[2]);

    uint32_t n_rays = 0, n_box = 0, n_tri = 0, n_shaded = 0, n_tex = 0, n_paths = 0, max_sp = 0;

    const int pass_end = pp.first_sample + pp.n_samples;
    int sample = pp.first_sample + chunk_id * pp.chunk;
    const int end = min(sample + pp.chunk, pass_end);
    bool alive = inside && (sample < end);

    int jitter_batch = -1;
    v3 prim_dir = mk3(0.0f, 0.0f, 1.0f);
    v3 color = mk3(1.0f, 1.0f, 1.0f);
    int depth = 0;
    RayD ray;
    ray_set(ray, cam_pos, prim_dir);

    auto begin_path = [&]() {
        const int bi = sample / pp.batch;
        if (bi != jitter_batch) {
            // one jittered primary ray per pixel per semantic batch (camera.cu:35-40, worker.cpp:71-78)
            jitter_batch = bi;
            uint32_t w0, w1, w2, w3;
            philox((uint32_t)pixel, (uint32_t)bi, 0u, 1u, k0, k1, w0, w1, w2, w3);
            prim_dir = primary_dir(pp.cam, x, y, u01(w0), u01(w1));
        }
        ray_set(ray, cam_pos, prim_dir);                 // raytrace.cu:210
        color = mk3(1.0f, 1.0f, 1.0f);                   // :211
        depth = pp.max_depth;                            // :212
    };

    if (alive) begin_path();

    while (alive) {
        bool finished = false, contributes = true;
        if (depth == 0) {
            // `while (depth--)` fell through: the path ran out of depth and adds nothing (raytrace.cu:243)
            finished = true; contributes = false;
        } else {
            depth--;
            Hit h;
            n_rays++;
            closest_hit<COUNT>(sc, ray, h, stack, RT_BLOCK, pp.lds_slots, n_box, n_tri, max_sp);
            if (h.tri == REF_NONE) {
                color = vmul(color, ambient); finished = true;      // raytrace.cu:220-223
            } else {
                if (COUNT) n_shaded++;
                // ---- triangle_interpolate, raytrace.cu:80-96 ----
                const float4* sq = reinterpret_cast<const float4*>(sc.shade + h.tri);
                const float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3];
                const float u = h.u, v = h.v;
                const float w = 1.0f - u - v;
                const v3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
                const v3 normal = vnormalize(vadd(vadd(vscale(n0, w), vscale(n1, u)), vscale(n2, v)));
                const int mat_id = __float_as_int(s3.w);
                const float4* mq = reinterpret_cast<const float4*>(sc.mats + mat_id);
                const float4 m0 = mq[0], m1 = mq[1];
                const int mtype = __float_as_int(m0.x);
                const float optical_density = m1.x, metal_fuzz = m1.y;
                const int mtex = __float_as_int(m1.z);
                v3 hit_color;
                if (mtex < 0) {
                    hit_color = mk3(m0.y, m0.z, m0.w);
                } else {
                    const float tcx = w * s2.y + u * s2.w + v * s3.y;
                    const float tcy = w * s2.z + u * s3.x + v * s3.z;
                    if (COUNT) n_tex++;
                    hit_color = tex_fetch(sc, mtex, tcx, tcy);
                }
                const v3 rd = mk3(ray.dx, ray.dy, ray.dz);
                const v3 point = vadd(mk3(ray.ox, ray.oy, ray.oz), vscale(rd, h.t));   // primitives.cuh:17-19
                const uint32_t bounce = (uint32_t)(pp.max_depth - 1 - depth);         // scatters done so far on this path
                uint32_t w_rr, w_dec, w2_, w3_;
                philox((uint32_t)pixel, (uint32_t)sample, bounce * 65536u, 0u, k0, k1, w_rr, w_dec, w2_, w3_);
                const float decision = u01(w_dec);
                if (mtype == RT_MAT_LIGHT) {
                    color = vmul(color, hit_color); finished = true;                 // raytrace.cu:226-229
                } else if (u01(w_rr) < pp.rr) {                                      // :230
                    // ---- material_scatter, raytrace.cu:120-195 ----
                    v3 dir = mk3(0.0f, 0.0f, 0.0f);
                    bool scattered = true;
                    const float dn = vdot(rd, normal);
                    if (mtype == RT_MAT_LAMBERTIAN) {
                        v3 d0 = normal;
                        if (dn > 0.0f) d0 = vneg(normal);
                        dir = vnormalize(vadd(d0, unit_sphere((uint32_t)pixel, (uint32_t)sample, bounce, k0, k1)));
                        const float cos_theta = fabsf(vdot(dir, normal));
                        color = vmul(color, vscale(hit_color, cos_theta));
                    } else if (mtype == RT_MAT_METAL) {
                        v3 nn = normal;
                        if (dn > 0.0f) nn = vneg(normal);
                        const v3 rnd = unit_sphere((uint32_t)pixel, (uint32_t)sample, bounce, k0, k1);
                        if (4.0f * decision < 3.0f + metal_fuzz) {
                            dir = vadd(nn, rnd);
                            const float cos_theta = fabsf(vdot(dir, nn));
                            color = vmul(color, vscale(hit_color, cos_theta));
                        } else {
                            dir = vadd(vnormalize(reflect3(rd, nn)), vscale(rnd, metal_fuzz));
                        }
                    } else if (mtype == RT_MAT_REFLECTIVE) {
                        dir = reflect3(rd, normal);
                        color = vmul(color, hit_color);
                    } else if (mtype == RT_MAT_REFRACTIVE) {
                        v3 outward;
                        float ni_over_nt, cosine;
                        if (dn > 0.0f) {
                            outward = vneg(normal);
                            ni_over_nt = optical_density;
                            cosine = optical_density * dn;
                        } else {
                            outward = normal;
                            ni_over_nt = 1.0f / optical_density;
                            cosine = -dn;
                        }
                        float reflect_prob = 1.0f;
                        {   // refract, raytrace.cu:102-110
                            const float dt = vdot(rd, outward);
                            const float disc = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
                            if (disc > 0.0f) {
                                dir = vsub(vscale(vsub(rd, vscale(outward, dt)), ni_over_nt), vscale(outward, sqrtf(disc)));
                                float r0 = (1.0f - optical_density) / (1.0f + optical_density);   // schlick, :112-116
                                r0 = r0 * r0;
                                reflect_prob = r0 + (1.0f - r0) * powi5(1.0f - cosine);
                            }
                        }
                        if (decision < reflect_prob) dir = reflect3(rd, normal);
                        color = vmul(color, hit_color);
                    } else {
                        scattered = false;
                    }
                    if (scattered) {
                        ray_set(ray, vadd(point, vscale(dir, 1e-4f)), dir);         // raytrace.cu:118
                    } else {
                        color = vmul(color, ambient); finished = true;             // :232-235
                    }
                } else {
                    color = vmul(color, ambient); finished = true;                 // :238-241
                }
            }
        }
        if (finished) {
            n_paths++;
            // a path that ran out of depth adds nothing (raytrace.cu:243): its slot holds +0, which is neutral
            // in the ordered sum because every other term is a product of non-negative factors
            if (!contributes) color = mk3(0.0f, 0.0f, 0.0f);
            float* slot = pp.slots + ((size_t)(sample - pp.first_sample) * (size_t)pp.n_pixel + (size_t)pixel) * 3;
            slot[0] = color.x; slot[1] = color.y; slot[2] = color.z;
            sample++;
            if (sample >= end) alive = false;
            else begin_path();
        }
    }

    wave_add(pp.counters + RT_CNT_RAYS, n_rays);
    wave_add(pp.counters + RT_CNT_PATHS, n_paths);
    if (COUNT) {
        wave_add(pp.counters + RT_CNT_BOX_TESTS, n_box);
        wave_add(pp.counters + RT_CNT_TRI_TESTS, n_tri);
        wave_add(pp.counters + RT_CNT_SHADED, n_shaded);
        wave_add(pp.counters + RT_CNT_TEX_FETCHES, n_tex);
        atomicMax(pp.counters + RT_CNT_MAX_STACK, (unsigned long long)max_sp);
    }
}

// Per pixel: sum the pass's slots in sample order into the running batch total and add every completed
// batch total to the output buffer -- the order of cuda/raytrace.cu:208-246 with the batch loop of
// csrc/worker.cpp:71-91 around it.  A batch that straddles two passes of one call carries its running
// total in `carry`; the last pass of a call flushes a final partial batch (the frame's remainder batch).
__global__ __launch_bounds__(RT_BLOCK) void accumulate_kernel(int n_pixel, int first_sample, int n_samples, int batch, int flush,
                                                              const float* __restrict__ slots, float* __restrict__ carry,
                                                              float* __restrict__ out) {
    const int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n_pixel) return;
    float ox = out[3 * (size_t)i], oy = out[3 * (size_t)i + 1], oz = out[3 * (size_t)i + 2];
    float tx = 0.0f, ty = 0.0f, tz = 0.0f;
    int in_batch = first_sample % batch;            // samples of the current batch summed by earlier passes
    if (in_batch) { tx = carry[3 * (size_t)i]; ty = carry[3 * (size_t)i + 1]; tz = carry[3 * (size_t)i + 2]; }
    for (int s = 0; s < n_samples; s++) {
        const float* p = slots + ((size_t)s * (size_t)n_pixel + (size_t)i) * 3;
        tx += p[0]; ty += p[1]; tz += p[2];
        if (++in_batch == batch) {
            ox += tx; oy += ty; oz += tz;
            tx = ty = tz = 0.0f;
            in_batch = 0;
        }
    }
    if (in_batch) {
        if (flush) { ox += tx; oy += ty; oz += tz; }
        else { carry[3 * (size_t)i] = tx; carry[3 * (size_t)i + 1] = ty; carry[3 * (size_t)i + 2] = tz; }
    }
    out[3 * (size_t)i] = ox; out[3 * (size_t)i + 1] = oy; out[3 * (size_t)i + 2] = oz;
}

// ---------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(RT_BLOCK) void closest_hit_kernel(SceneDev sc, int lds_slots, int n_rays, const float* __restrict__ rays,
                                                               int32_t* obj, int32_t* tri, float* t, float* u, float* v,
                                                               unsigned long long* counters) {
    extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];
    lds_uint2* stack = (lds_uint2*)lds_stack + threadIdx.x;
    const int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    uint32_t n_box = 0, n_tri = 0, n_r = 0, max_sp = 0;
    if (i < n_rays) {
        RayD r;
        ray_set(r, mk3(rays[6 * (size_t)i], rays[6 * (size_t)i + 1], rays[6 * (size_t)i + 2]),
                mk3(rays[6 * (size_t)i + 3], rays[6 * (size_t)i + 4], rays[6 * (size_t)i + 5]));
        Hit h;
        n_r = 1;
        closest_hit<true>(sc, r, h, stack, RT_BLOCK, lds_slots, n_box, n_tri, max_sp);
        int o = -1, ti = -1;
        if (h.tri != REF_NONE) { const TriOrigin og = sc.origin[h.tri]; o = og.object; ti = og.triangle; }
        obj[i] = o; tri[i] = ti; t[i] = h.t;
        if (u) u[i] = h.u;
        if (v) v[i] = h.v;
    }
    wave_add(counters + RT_CNT_RAYS, n_r);
    wave_add(counters + RT_CNT_BOX_TESTS, n_box);
    wave_add(counters + RT_CNT_TRI_TESTS, n_tri);
    atomicMax(counters + RT_CNT_MAX_STACK, (unsigned long long)max_sp);
}

__global__ __launch_bounds__(RT_BLOCK) void primary_rays_kernel(CameraFrame cam, uint32_t key0, uint32_t key1, uint32_t batch_index,
                                                                float* __restrict__ rays) {
    const int idx = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (idx >= cam.width * cam.height) return;
    const int x = idx % cam.width, y = idx / cam.width;
    uint32_t w0, w1, w2, w3;
    philox((uint32_t)idx, batch_index, 0u, 1u, key0, key1, w0, w1, w2, w3);
    const v3 d = primary_dir(cam, x, y, u01(w0), u01(w1));
    float* o = rays + 6 * (size_t)idx;
    o[0] = cam.pos[0]; o[1] = cam.pos[1]; o[2] = cam.pos[2];
    o[3] = d.x; o[4] = d.y; o[5] = d.z;
}

__global__ __launch_bounds__(RT_BLOCK) void post_process_kernel(int n_pixels, int spp, float gamma, float* __restrict__ buf) {   // raytrace.cu:249-261
    const int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n_pixels) return;
    const float inv_spp = 1.f / (float)spp;          // `/` on float3 multiplies by the reciprocal (common.cuh:148-153)
    const float inv_gamma = 1.0f / gamma;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float val = buf[3 * (size_t)i + c] * inv_spp;
        val = fminf(1.0f, fmaxf(0.0f, val));
        buf[3 * (size_t)i + c] = powf(val, inv_gamma);
    }
}

// ---------------------------------------------------------------------------
// context and C ABI
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) { g_last_error = msg; return code; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct rt_context {
    int device = 0;
    SceneDev dev{};
    void* d_nodes = nullptr; void* d_isect = nullptr; void* d_shade = nullptr;
    void* d_origin = nullptr; void* d_mats = nullptr; void* d_tex = nullptr; void* d_texels = nullptr;
    unsigned long long* d_counters = nullptr;
    bool have_scene = false, have_camera = false;
    CameraFrame cam{};
    uint64_t seed = 0;
    int width = 0, height = 0;
    int opt_counters = 0, opt_xcd = 1;
    int opt_lds_slots = 12;          // stack levels kept in LDS (deeper ones spill to private memory)
    int opt_chunk = 8;               // samples one lane traces in a row
    int opt_pass = 256;              // samples per pixel held in the slot buffer at once
    float* d_slots = nullptr;        // [pass][n_pixel][3], then one [n_pixel][3] carry plane
    size_t slots_bytes = 0;
    int stack_slots = 1;
    uint64_t stats[8] = {0};
};

static void free_scene(rt_context* c) {
    void** ptrs[] = {&c->d_nodes, &c->d_isect, &c->d_shade, &c->d_origin, &c->d_mats, &c->d_tex, &c->d_texels};
    for (void** p : ptrs) { if (*p) { (void)hipFree(*p); *p = nullptr; } }
    c->have_scene = false;
}

template <class T>
static int upload_vec(const std::vector<T>& v, void** d, uint64_t& bytes_total) {
    const size_t bytes = std::max<size_t>(v.size() * sizeof(T), 64);
    HIP_TRY(hipMalloc(d, bytes));
    if (!v.empty()) HIP_TRY(hipMemcpy(*d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    bytes_total += bytes;
    return RT_OK;
}

extern "C" {

const char* rt_last_error(void) { return g_last_error.c_str(); }
int rt_abi_version(void) { return RTCORE_ABI_VERSION; }

int rt_build_bvh(const rt_triangle* triangles, int n_triangles, rt_bvh_node* nodes_out, int capacity) {
    if (!triangles || !nodes_out || n_triangles <= 0) return fail(RT_ERR_INVALID, "rt_build_bvh: null array or n_triangles <= 0");
    std::vector<rt_bvh_node> nodes;
    const int n = build_bvh(triangles, n_triangles, nodes);
    if (n > capacity) return fail(RT_ERR_CAPACITY, "rt_build_bvh: need " + std::to_string(n) + " nodes");
    std::memcpy(nodes_out, nodes.data(), (size_t)n * sizeof(rt_bvh_node));
    return n;
}

int rt_apply_transform(const rt_transform* transform, rt_triangle* triangles, int n_triangles) {
    if (!transform || (!triangles && n_triangles > 0) || n_triangles < 0) return fail(RT_ERR_INVALID, "rt_apply_transform: bad argument");
    apply_transform(*transform, triangles, n_triangles);
    return RT_OK;
}

int rt_create(int device_id, rt_context** out) {
    if (!out) return fail(RT_ERR_INVALID, "rt_create: null out");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(RT_ERR_INVALID, "rt_create: no such device");
    HIP_TRY(hipSetDevice(device_id));
    rt_context* c = new rt_context();
    c->device = device_id;
    hipError_t e = hipMalloc((void**)&c->d_counters, RT_CNT_COUNT * sizeof(unsigned long long));
    if (e != hipSuccess) { delete c; return fail(RT_ERR_DEVICE, std::string("hipMalloc counters: ") + hipGetErrorString(e)); }
    (void)hipMemset(c->d_counters, 0, RT_CNT_COUNT * sizeof(unsigned long long));
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_context* ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_scene(ctx);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    delete ctx;
    return RT_OK;
}

int rt_set_option(rt_context* ctx, int option, int value) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_set_option: null context");
    switch (option) {
    case RT_OPT_COUNTERS: ctx->opt_counters = value ? 1 : 0; return RT_OK;
    case RT_OPT_XCD_REMAP: ctx->opt_xcd = value ? 1 : 0; return RT_OK;
    case RT_OPT_LDS_SLOTS:
        if (value < 1 || value > 32) return fail(RT_ERR_INVALID, "rt_set_option: LDS stack slots must be 1..32");
        ctx->opt_lds_slots = value; return RT_OK;
    case RT_OPT_CHUNK:
        if (value < 1 || value > 65536) return fail(RT_ERR_INVALID, "rt_set_option: chunk must be 1..65536");
        ctx->opt_chunk = value; return RT_OK;
    case RT_OPT_PASS:
        if (value < 1 || value > 65536) return fail(RT_ERR_INVALID, "rt_set_option: pass must be 1..65536");
        ctx->opt_pass = value; return RT_OK;
    default: return fail(RT_ERR_INVALID, "rt_set_option: unknown option");
    }
}

int rt_upload_scene(rt_context* ctx, const rt_object* objects, int n_objects, const rt_material* materials, int n_materials,
                    const rt_texture* textures, int n_textures) {
    if (!ctx || !objects || n_objects <= 0 || !materials || n_materials <= 0 || n_textures < 0 || (n_textures > 0 && !textures))
        return fail(RT_ERR_INVALID, "rt_upload_scene: bad argument");
    // every material id and texture index must be in range: the kernels index with them unchecked
    for (int i = 0; i < n_objects; i++) {
        if (!objects[i].triangles || objects[i].n_triangles <= 0) return fail(RT_ERR_INVALID, "rt_upload_scene: object without triangles");
        for (int k = 0; k < objects[i].n_triangles; k++) {
            const int m = objects[i].triangles[k].material_id;
            if (m < 0 || m >= n_materials) return fail(RT_ERR_INVALID, "rt_upload_scene: material id out of range");
        }
    }
    for (int i = 0; i < n_materials; i++)
        if (materials[i].texture >= n_textures) return fail(RT_ERR_INVALID, "rt_upload_scene: texture index out of range");
    for (int i = 0; i < n_textures; i++)
        if (!textures[i].rgba || textures[i].width <= 0 || textures[i].height <= 0) return fail(RT_ERR_INVALID, "rt_upload_scene: empty texture");

    PackedScene ps;
    std::string err;
    if (pack_scene(objects, n_objects, ps, err) != 0) return fail(RT_ERR_INVALID, "rt_upload_scene: " + err);

    std::vector<MatDev> mats(n_materials);
    for (int i = 0; i < n_materials; i++) {
        mats[i].type = materials[i].type;
        mats[i].albedo[0] = materials[i].albedo.x; mats[i].albedo[1] = materials[i].albedo.y; mats[i].albedo[2] = materials[i].albedo.z;
        mats[i].optical_density = materials[i].optical_density;
        mats[i].metal_fuzz = materials[i].metal_fuzz;
        mats[i].texture = materials[i].texture < 0 ? -1 : materials[i].texture;
        mats[i].pad = 0;
    }
    std::vector<TexDev> tex(n_textures);
    std::vector<uint32_t> texels;
    for (int i = 0; i < n_textures; i++) {
        tex[i].offset = (uint32_t)texels.size();
        tex[i].width = textures[i].width; tex[i].height = textures[i].height; tex[i].pad = 0;
        const size_t n = (size_t)textures[i].width * textures[i].height;
        const size_t at = texels.size();
        texels.resize(at + n);
        std::memcpy(texels.data() + at, textures[i].rgba, n * 4);
    }

    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    free_scene(ctx);
    uint64_t bytes = 0;
    int rc;
    if ((rc = upload_vec(ps.nodes, &ctx->d_nodes, bytes))) return rc;
    if ((rc = upload_vec(ps.isect, &ctx->d_isect, bytes))) return rc;
    if ((rc = upload_vec(ps.shade, &ctx->d_shade, bytes))) return rc;
    if ((rc = upload_vec(ps.origin, &ctx->d_origin, bytes))) return rc;
    if ((rc = upload_vec(mats, &ctx->d_mats, bytes))) return rc;
    if ((rc = upload_vec(tex, &ctx->d_tex, bytes))) return rc;
    if ((rc = upload_vec(texels, &ctx->d_texels, bytes))) return rc;
    ctx->dev.root = ps.root;
    ctx->dev.nodes = (const Node64*)ctx->d_nodes;
    ctx->dev.isect = (const TriIsect*)ctx->d_isect;
    ctx->dev.shade = (const TriShade*)ctx->d_shade;
    ctx->dev.origin = (const TriOrigin*)ctx->d_origin;
    ctx->dev.mats = (const MatDev*)ctx->d_mats;
    ctx->dev.tex = (const TexDev*)ctx->d_tex;
    ctx->dev.texels = (const uint32_t*)ctx->d_texels;
    ctx->dev.n_materials = n_materials;
    ctx->stack_slots = ps.max_depth > 0 ? ps.max_depth : 1;
    ctx->stats[0] = ps.isect.size();
    ctx->stats[1] = ps.nodes.size();
    ctx->stats[2] = (uint64_t)ctx->stack_slots;
    ctx->stats[3] = bytes;
    ctx->stats[4] = (uint64_t)n_objects;
    ctx->have_scene = true;
    return RT_OK;
}

int rt_init_randstate(rt_context* ctx, uint64_t seed, int width, int height) {
    if (!ctx || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "rt_init_randstate: bad argument");
    ctx->seed = seed;
    ctx->width = width;
    ctx->height = height;
    return RT_OK;
}

int rt_setup_raytrace(rt_context* ctx, int width, int height, const float camera_pos[3], const float camera_dir[3],
                      const float camera_up[3], float fov_radians) {
    if (!ctx || width <= 0 || height <= 0 || !camera_pos || !camera_dir || !camera_up)
        return fail(RT_ERR_INVALID, "rt_setup_raytrace: bad argument");
    if ((long long)width * height > 0x7FFFFFFFLL / 6) return fail(RT_ERR_INVALID, "rt_setup_raytrace: frame too large");
    ctx->cam = camera_frame(width, height, camera_pos, camera_dir, camera_up, fov_radians);
    ctx->width = width;
    ctx->height = height;
    ctx->have_camera = true;
    return RT_OK;
}

static int lds_slots(const rt_context* ctx) { return ctx->stack_slots < ctx->opt_lds_slots ? ctx->stack_slots : ctx->opt_lds_slots; }
static size_t stack_bytes(const rt_context* ctx) { return (size_t)lds_slots(ctx) * RT_BLOCK * sizeof(uint2); }
static bool stack_fits(const rt_context* ctx) { return ctx->stack_slots <= lds_slots(ctx) + RT_SPILL_SLOTS; }

int rt_raytrace(rt_context* ctx, int n_pixel, int first_sample, int spp, int semantic_batch, int max_depth, const float ambient[3],
                float russian_roulette, float* d_output, void* stream) {
    if (!ctx || !ambient || !d_output) return fail(RT_ERR_INVALID, "rt_raytrace: null argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "rt_raytrace: no scene uploaded");
    if (!ctx->have_camera) return fail(RT_ERR_NO_CAMERA, "rt_raytrace: rt_setup_raytrace not called");
    if (n_pixel != ctx->cam.width * ctx->cam.height) return fail(RT_ERR_INVALID, "rt_raytrace: n_pixel does not match the camera frame");
    if (spp < 0 || first_sample < 0 || semantic_batch <= 0 || max_depth < 0) return fail(RT_ERR_INVALID, "rt_raytrace: negative count");
    if (first_sample % semantic_batch) return fail(RT_ERR_INVALID, "rt_raytrace: first_sample must be a multiple of semantic_batch");
    if ((long long)first_sample + spp > 0x7FFFFFFFLL - semantic_batch) return fail(RT_ERR_INVALID, "rt_raytrace: sample index overflow");
    if (spp == 0) return RT_OK;
    if (!stack_fits(ctx)) return fail(RT_ERR_INVALID, "rt_raytrace: BVH deeper than the traversal stack allows");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    // slot buffer: one float3 per (sample of the pass, pixel); grown on demand, kept by the context
    const int pass = spp < ctx->opt_pass ? spp : ctx->opt_pass;
    const size_t plane = (size_t)n_pixel * 3 * sizeof(float);
    const size_t need = ((size_t)pass + 1) * plane;
    if (need > ctx->slots_bytes) {
        HIP_TRY(hipStreamSynchronize(s));
        if (ctx->d_slots) { HIP_TRY(hipFree(ctx->d_slots)); ctx->d_slots = nullptr; ctx->slots_bytes = 0; }
        HIP_TRY(hipMalloc((void**)&ctx->d_slots, need));
        ctx->slots_bytes = need;
    }
    PathParams pp;
    pp.cam = ctx->cam;
    pp.n_pixel = n_pixel;
    pp.batch = semantic_batch;
    pp.max_depth = max_depth;
    pp.ambient[0] = ambient[0]; pp.ambient[1] = ambient[1]; pp.ambient[2] = ambient[2];
    pp.rr = russian_roulette;
    pp.key0 = (uint32_t)ctx->seed; pp.key1 = (uint32_t)(ctx->seed >> 32);
    pp.slots = ctx->d_slots;
    pp.counters = ctx->d_counters;
    pp.tiles_x = (ctx->cam.width + 15) / 16;
    pp.tiles_y = (ctx->cam.height + 15) / 16;
    pp.xcd_remap = ctx->opt_xcd;
    pp.chunk = ctx->opt_chunk;
    pp.lds_slots = lds_slots(ctx);
    for (int done = 0; done < spp; done += pass) {
        pp.first_sample = first_sample + done;
        pp.n_samples = spp - done < pass ? spp - done : pass;
        pp.n_chunks = (pp.n_samples + pp.chunk - 1) / pp.chunk;
        const long long nblocks = (long long)pp.tiles_x * pp.tiles_y * pp.n_chunks;
        if (nblocks > 0x7FFFFFFFLL) return fail(RT_ERR_INVALID, "rt_raytrace: grid too large");
        if (ctx->opt_counters)
            hipLaunchKernelGGL(path_kernel<true>, dim3((unsigned)nblocks), dim3(RT_BLOCK), stack_bytes(ctx), s, ctx->dev, pp);
        else
            hipLaunchKernelGGL(path_kernel<false>, dim3((unsigned)nblocks), dim3(RT_BLOCK), stack_bytes(ctx), s, ctx->dev, pp);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(accumulate_kernel, dim3((n_pixel + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, s, n_pixel, pp.first_sample,
                           pp.n_samples, semantic_batch, done + pass >= spp ? 1 : 0, (const float*)ctx->d_slots,
                           ctx->d_slots + (size_t)pass * (size_t)n_pixel * 3, d_output);
        HIP_TRY(hipGetLastError());
    }
    return RT_OK;
}

int rt_post_process(rt_context* ctx, int n_pixels, int spp, float gamma, float* d_output, void* stream) {
    if (!ctx || !d_output || n_pixels < 0 || spp <= 0) return fail(RT_ERR_INVALID, "rt_post_process: bad argument");
    if (n_pixels == 0) return RT_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(post_process_kernel, dim3((n_pixels + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, (hipStream_t)stream,
                       n_pixels, spp, gamma, d_output);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_primary_rays(rt_context* ctx, uint32_t batch_index, float* d_rays, void* stream) {
    if (!ctx || !d_rays) return fail(RT_ERR_INVALID, "rt_primary_rays: null argument");
    if (!ctx->have_camera) return fail(RT_ERR_NO_CAMERA, "rt_primary_rays: rt_setup_raytrace not called");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = ctx->cam.width * ctx->cam.height;
    hipLaunchKernelGGL(primary_rays_kernel, dim3((n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, (hipStream_t)stream,
                       ctx->cam, (uint32_t)ctx->seed, (uint32_t)(ctx->seed >> 32), batch_index, d_rays);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_closest_hit(rt_context* ctx, int n_rays, const float* d_rays, int32_t* d_object, int32_t* d_triangle, float* d_t, float* d_u,
                   float* d_v, void* stream) {
    if (!ctx || n_rays < 0 || (n_rays > 0 && (!d_rays || !d_object || !d_triangle || !d_t)))
        return fail(RT_ERR_INVALID, "rt_closest_hit: bad argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "rt_closest_hit: no scene uploaded");
    if (n_rays == 0) return RT_OK;
    if (!stack_fits(ctx)) return fail(RT_ERR_INVALID, "rt_closest_hit: BVH deeper than the traversal stack allows");
    HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(closest_hit_kernel, dim3((n_rays + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), stack_bytes(ctx), (hipStream_t)stream,
                       ctx->dev, lds_slots(ctx), n_rays, d_rays, d_object, d_triangle, d_t, d_u, d_v, ctx->d_counters);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_get_counters(rt_context* ctx, uint64_t out[RT_CNT_COUNT]) {
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_get_counters: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, ctx->d_counters, RT_CNT_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_reset_counters(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_reset_counters: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(ctx->d_counters, 0, RT_CNT_COUNT * sizeof(uint64_t)));
    return RT_OK;
}

int rt_scene_stats(rt_context* ctx, uint64_t out[8]) {
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_scene_stats: null argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "rt_scene_stats: no scene uploaded");
    std::memcpy(out, ctx->stats, sizeof(ctx->stats));
    return RT_OK;
}

int rt_device_alloc(rt_context* ctx, uint64_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr) return fail(RT_ERR_INVALID, "rt_device_alloc: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
    return RT_OK;
}

int rt_device_free(rt_context* ctx, void* d_ptr) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_device_free: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return RT_OK;
}

int rt_device_zero(rt_context* ctx, void* d_ptr, uint64_t bytes, void* stream) {
    if (!ctx || (!d_ptr && bytes)) return fail(RT_ERR_INVALID, "rt_device_zero: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes) HIP_TRY(hipMemsetAsync(d_ptr, 0, bytes, (hipStream_t)stream));
    return RT_OK;
}

int rt_copy_to_host(rt_context* ctx, void* h_dst, const void* d_src, uint64_t bytes) {
    if (!ctx || ((!h_dst || !d_src) && bytes)) return fail(RT_ERR_INVALID, "rt_copy_to_host: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes) HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_copy_to_device(rt_context* ctx, void* d_dst, const void* h_src, uint64_t bytes) {
    if (!ctx || ((!d_dst || !h_src) && bytes)) return fail(RT_ERR_INVALID, "rt_copy_to_device: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes) HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}

int rt_synchronize(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_synchronize: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    return RT_OK;
}

}  // extern "C"
