// rt_device.h -- device-side building blocks shared by every kernel: float3 math in upstream's operand
// order, the addressed Philox generator, the slab / Moller-Trumbore tests, the closest-hit walk and the
// shading step.  Compiled with -ffp-contract=off.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>

#include "rt_host.h"

using namespace rtc;

// ---------------------------------------------------------------------------
// device-side scene view and parameters
// ---------------------------------------------------------------------------
struct SceneDev {
    ChildRec root;            // box + reference of the whole scene (root of the object-level tree)
    const Node4* nodes;
    const TriIsect* isect;
    const TriShade* shade;
    const TriOrigin* origin;
    const MatDev* mats;
    const TexDev* tex;
    const uint32_t* texels;
    int n_materials;
    uint32_t last_tri;        // the triangle a walk that enters every box tests last (REF_NONE in an empty scene)
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
    int sort_rays;                        // unused (the compaction no longer orders by direction octant)
    int hist_bands;                       // calibration pass only: the histogram has this many parts, one per band of tile rows (0: one part)
    uint32_t* hist;                       // calibration pass only: per-node visit counts (see TravCounters), else null
    // wavefront: divisors the refill and the shading kernel divide path numbers by, with floor(2^32 / divisor) beside them
    uint32_t per_tile, per_tile_magic;    // pool entries one 16x16 tile starts in this pass = n_runs * 256
    uint32_t tiles_x_magic, batch_magic;
    // wavefront: a new pool entry stands for a RUN of samples of one pixel that share their primary ray and therefore their
    // first hit (cuda/raytrace.cu:210: every sample of a semantic batch starts with the batch's one jittered ray): run k of
    // a pass covers the samples of [(run_base + k) * run_len, (run_base + k + 1) * run_len) that lie in the pass.
    // run_len = the semantic batch (one closest-hit query per pixel and batch), or 1 (one per sample, as upstream runs it).
    int run_len, run_base, n_runs;
    int run_lanes;                        // lanes the shading launch gives a new entry: min(run_len, n_samples)
    uint32_t run_lanes_magic;
    // Where a path's radiance goes.  ring_tiles == 0: slots[sample of the pass][pixel] (one launch traces the whole
    // pass, the ordered sum runs after it).  ring_tiles > 0 (wavefront): slots[tile % ring_tiles][sample of the
    // pass][pixel of the 16x16 tile] -- the pass starts its paths tile after tile, so a tile's slots can be summed and
    // handed on to a later tile as soon as its last path has ended, and the buffer holds ring_tiles tiles, not the frame.
    uint32_t ring_tiles, ring_magic, width_magic;
    // Order of a tile's entries when every sample has an entry of its own (run_len == 1).  group_log2 == 0: run after run,
    // each a tile's 256 pixels.  Otherwise runs come in groups of G = 2^group_log2 and a group is laid out pixel after
    // pixel, each pixel's G samples side by side: the samples of a semantic batch start with one and the same primary ray
    // (cuda/raytrace.cu:210), and rays that coincide, held by neighbouring lanes, ask the L1 for one address where
    // scattered lanes ask for G.  n_full_groups whole groups, then one of group_rem runs.
    uint32_t group_log2, n_full_groups, group_rem, group_rem_magic;
};

// x / d for a divisor the whole launch shares (see udiv_by below)
__device__ __forceinline__ uint32_t udiv_by(uint32_t x, uint32_t d, uint32_t m, uint32_t& rem);

// the slot of (pixel, sample) in the layout the pass uses
__device__ __forceinline__ float* slot_of(const PathParams& pp, uint32_t pixel, int sample) {
    if (pp.ring_tiles == 0u) return pp.slots + ((size_t)(sample - pp.first_sample) * (size_t)pp.n_pixel + (size_t)pixel) * 3;
    uint32_t x, place;
    const uint32_t y = udiv_by(pixel, (uint32_t)pp.cam.width, pp.width_magic, x);
    const uint32_t tile = (y >> 4) * (uint32_t)pp.tiles_x + (x >> 4);
    (void)udiv_by(tile, pp.ring_tiles, pp.ring_magic, place);
    const uint32_t in_tile = ((y & 15u) << 4) | (x & 15u);
    return pp.slots + (((size_t)place * (size_t)pp.n_samples + (size_t)(sample - pp.first_sample)) * 256u + in_tile) * 3;
}

// samples [lo, hi) of run k of the pass (see PathParams)
__device__ __forceinline__ void run_samples(const PathParams& pp, uint32_t k, int& lo, int& hi) {
    const int a = (pp.run_base + (int)k) * pp.run_len, b = a + pp.run_len, end = pp.first_sample + pp.n_samples;
    lo = a > pp.first_sample ? a : pp.first_sample;
    hi = b < end ? b : end;
}

// x / d for a divisor the whole launch shares, with m = floor(2^32 / d) from the host: the estimate mulhi(x, m) is the
// quotient or one below it (two corrections allowed for), which costs a multiply instead of the ~35 instructions of a
// 32-bit division
__device__ __forceinline__ uint32_t udiv_by(uint32_t x, uint32_t d, uint32_t m, uint32_t& rem) {
    uint32_t q = __umulhi(x, m);
    uint32_t r = x - q * d;
    if (r >= d) { q++; r -= d; }
    if (r >= d) { q++; r -= d; }
    rem = r;
    return q;
}

// place of (run k, pixel p of the tile) among the tile's entries, and the run of a place (see PathParams::group_log2)
__device__ __forceinline__ uint32_t place_of(const PathParams& pp, uint32_t k, uint32_t p) {
    if (pp.group_log2 == 0u) return k * 256u + p;
    const uint32_t g = k >> pp.group_log2, span = 8u + pp.group_log2;
    if (g < pp.n_full_groups) return (g << span) + (p << pp.group_log2) + (k & ((1u << pp.group_log2) - 1u));
    return (pp.n_full_groups << span) + p * pp.group_rem + (k - (pp.n_full_groups << pp.group_log2));
}
__device__ __forceinline__ uint32_t run_of_place(const PathParams& pp, uint32_t w) {
    if (pp.group_log2 == 0u) return w >> 8;
    const uint32_t span = 8u + pp.group_log2, g = w >> span;
    if (g < pp.n_full_groups) return (g << pp.group_log2) + (w & ((1u << pp.group_log2) - 1u));
    uint32_t j;
    (void)udiv_by(w - (pp.n_full_groups << span), pp.group_rem, pp.group_rem_magic, j);
    return (pp.n_full_groups << pp.group_log2) + j;
}

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
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a high and a low multiply
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
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
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

struct RayD {
    float ox, oy, oz;
    float dx, dy, dz;
    float ix, iy, iz;    // 1/d, hoisted out of include/bvh.h:43
    // the same origin and inverses again as register pairs in the order a box record holds its planes
    // (min.x min.y | min.z max.x | max.y max.z), so that one packed instruction handles two planes
    f2v o_xy, o_zx, o_yz, i_xy, i_zx, i_yz;
};

__device__ __forceinline__ void ray_set(RayD& r, v3 o, v3 d) {
    r.ox = o.x; r.oy = o.y; r.oz = o.z;
    r.dx = d.x; r.dy = d.y; r.dz = d.z;
    r.ix = 1.0f / d.x; r.iy = 1.0f / d.y; r.iz = 1.0f / d.z;
    r.o_xy = (f2v){r.ox, r.oy}; r.o_zx = (f2v){r.oz, r.ox}; r.o_yz = (f2v){r.oy, r.oz};
    r.i_xy = (f2v){r.ix, r.iy}; r.i_zx = (f2v){r.iz, r.ix}; r.i_yz = (f2v){r.iy, r.iz};
}

// include/bvh.h:39-58 on a box as it lies in memory: three pairs (min.x min.y | min.z max.x | max.y max.z).
// The per-axis early exit there is equivalent to one final t0 > t1 test because t0 never decreases and t1
// never increases.  The six subtractions and six multiplications run as three packed pairs each -- lane for
// lane the same IEEE operations as the scalar form.
// Upstream starts from t0 = -inf, t1 = +inf and folds the axes in with fmaxf / fminf, which drop a NaN operand (0 * inf
// on an axis the ray runs parallel to, with its origin on the plane).  Here the three axes go into one three-operand
// maximum and minimum -- the same value unless ALL three operands are NaN, where upstream keeps -inf / +inf and this
// keeps the NaN.  Everything that reads the results is written so that the two cases cannot be told apart:
//   t0 is only ever the left side of a "not greater than" test (against t1, against the best t, at once or when it
//      comes off the stack): false for NaN, false for -inf;
//   t1 is the right side of the same test against t0 (t0 > +inf and t0 > NaN are both false) and must be positive:
//      asked as "not (t1 <= 0)", which is true for NaN as t1 > 0 is for +inf, and the same thing for every number.
__device__ __forceinline__ bool slab(f2v p0, f2v p1, f2v p2, const RayD& r, float& t_near) {
    const f2v d0 = (p0 - r.o_xy) * r.i_xy;      // ax, ay
    const f2v d1 = (p1 - r.o_zx) * r.i_zx;      // az, bx
    const f2v d2 = (p2 - r.o_yz) * r.i_yz;      // by, bz
    const float ax = d0.x, ay = d0.y, az = d1.x, bx = d1.y, by = d2.x, bz = d2.y;
    const bool nx = r.i_xy.x < 0.0f, ny = r.i_xy.y < 0.0f, nz = r.i_zx.x < 0.0f;
    const float t0 = fmaxf(fmaxf(nx ? bx : ax, ny ? by : ay), nz ? bz : az);
    const float t1 = fminf(fminf(nx ? ax : bx, ny ? ay : by), nz ? az : bz);
    t_near = t0;
    return !(t0 > t1) & !(t1 <= 0.0f);
}
__device__ __forceinline__ bool slab(f4v a, f4v b, const RayD& r, float& t_near) { return slab(a.xy, a.zw, b.xy, r, t_near); }

__device__ __forceinline__ float4 as_float4(f4v v) { return make_float4(v.x, v.y, v.z, v.w); }

struct Hit {
    float t, u, v;
    uint32_t tri;      // packed triangle index, REF_NONE when nothing was hit
};

// cuda/raytrace.cu:24-42 on a packed record (e1, e2 precomputed), already in registers.  Rejections keep
// upstream's comparison polarity so that NaNs fall the same way.
__device__ __forceinline__ void tri_test(float4 q0, float4 q1, float4 q2, uint32_t p, const RayD& r, Hit& h) {
    const float v0x = q0.x, v0y = q0.y, v0z = q0.z;
    const float e1x = q0.w, e1y = q1.x, e1z = q1.y;
    const float e2x = q1.z, e2y = q1.w, e2z = q2.x;
    const float px = r.dy * e2z - r.dz * e2y;
    const float py = r.dz * e2x - r.dx * e2z;
    const float pz = r.dx * e2y - r.dy * e2x;
    const float det = e1x * px + e1y * py + e1z * pz;
    bool rej = (det > -1e-6f) && (det < 1e-6f);
    const float inv_det = 1.0f / det;
    const float tx = r.o_xy.x + (-v0x), ty = r.o_xy.y + (-v0y), tz = r.o_zx.x + (-v0z);   // the traversal keeps the origin in its pairs
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
// The walk runs on the two-level nodes (rt_packed.h, Node4).
// `stack` points at this lane's column of the LDS stack; consecutive slots are `stride` entries apart.
// The first `lds_slots` stack levels live in LDS; deeper levels (rare: a level is only used while BOTH
// children of a node are hit) spill to a per-lane private array, so LDS no longer caps occupancy.
#define RT_SPILL_SLOTS 56
typedef __attribute__((address_space(3))) uint2 lds_uint2;   // explicitly LDS: never merged with the private spill into a flat access
typedef __attribute__((address_space(3))) f4v lds_float4;   // native vector type: HIP's float4 class cannot be assigned across address spaces
// LDS copies of hot nodes hold the 112 bytes a visit reads, back to back: consecutive nodes start 28 banks further
// on, so lanes that read the same 16-byte piece of different nodes spread over the banks.
#define RT_HOT_STRIDE 112

#define RT_HIST_NODES 4096u          // node records whose visits the calibration pass counts one by one (the candidates for the LDS copy)
struct TravCounters {
    uint32_t n_box, n_tri, max_sp, w_node, w_leaf, n_node, n_hot_visit, ray_visits, max_ray_visits, n_nonfinite;
    uint32_t* hist;                  // counting builds only: visits per node record below RT_HIST_NODES, or null
};

// The walk, one step at a time, so that kernels can interleave it with other work (refilling idle lanes).
// State: `cur` = node or leaf to visit next (REF_NONE = walk over), `sp` = stack height.
template <bool COUNT>
struct Walker {
    uint32_t cur;
    int sp;
    lds_uint2* stack;
    int stride, lds_slots;
    lds_float4* hot;
    uint32_t n_hot;
    // The first `top_n` records may be read from a renumbered copy of their own at `top` (numbers are only addresses;
    // `root_ref` is the root's number in that copy): the copy whose first n_hot records are the ones the rays of the part
    // of the picture now in flight enter most.  top_n == 0: one array, the scene's.
    const Node4* top;
    uint32_t top_n, root_ref;
    uint2* spill;                  // RT_SPILL_SLOTS private entries owned by the caller (kept out of this struct so that its scalars stay in registers)

    __device__ __forceinline__ void init(lds_uint2* stack_, int stride_, int lds_slots_, uint2* spill_) {
        stack = stack_; stride = stride_; lds_slots = lds_slots_; spill = spill_; cur = REF_NONE; sp = 0;
        hot = nullptr; n_hot = 0;
        top = nullptr; top_n = 0; root_ref = REF_NONE;
    }

    // The first `n_hot_` node records (breadth-first numbering: the ones nearest to the root, which take more than
    // half of all visits) have been copied to LDS at `hot_`, RT_HOT_STRIDE bytes apart.
    __device__ __forceinline__ void set_hot(lds_float4* hot_, uint32_t n_hot_) { hot = hot_; n_hot = n_hot_; }
    __device__ __forceinline__ void set_top(const Node4* top_, uint32_t top_n_, uint32_t root_ref_) { top = top_; top_n = top_n_; root_ref = root_ref_; }

    __device__ __forceinline__ void begin(const SceneDev& sc, const RayD& r, Hit& h, TravCounters& tc) {
        sp = 0;
        cur = first_ref(sc, r, h, tc);
    }

    // Start of the walk for ray `r`: resets `h` and returns what the walk visits first (REF_NONE: nothing).  Touches no
    // walker state, so a caller can keep its own walk going in the lanes that take no new ray.
    __device__ __forceinline__ uint32_t first_ref(const SceneDev& sc, const RayD& r, Hit& h, TravCounters& tc) const {
        h.t = RT_INF; h.u = 0.0f; h.v = 0.0f; h.tri = REF_NONE;
        uint32_t first = REF_NONE;
        float tn;
        if (COUNT) {
            tc.n_box++;
            tc.max_ray_visits = max(tc.max_ray_visits, tc.ray_visits);
            tc.ray_visits = 0;
            const float sum = r.dx + r.dy + r.dz;
            tc.n_nonfinite += (sum - sum == 0.0f) ? 0u : 1u;      // a NaN or infinite direction component
        }
        const f4v a = {sc.root.bmin[0], sc.root.bmin[1], sc.root.bmin[2], sc.root.bmax[0]};
        const f4v b = {sc.root.bmax[1], sc.root.bmax[2], 0.0f, 0.0f};
        // A ray whose direction is exactly zero passes the slab test of every box beyond its origin, so upstream's walk
        // visits a large part of the tree -- and accepts nothing: d x e2 = 0 makes det = +-0 for every finite triangle,
        // which the |det| < 1e-6 test rejects (raytrace.cu:27-29).  The answer "no hit" is given without the walk.
        const bool still = r.dx == 0.0f && r.dy == 0.0f && r.dz == 0.0f;
        // A ray whose three direction components are NaN enters every box (NaN products drop out of the slab test) and
        // passes every triangle test (det is NaN, every rejection compares false; raytrace.cu:27-41), each acceptance
        // overwriting the one before: upstream's answer is the triangle its walk tests last, with that test's NaN
        // t, u, v.  That one test is done here; the walk over the whole tree is not.
        const bool lost = (r.dx != r.dx) && (r.dy != r.dy) && (r.dz != r.dz);
        if (lost) {
            if (sc.last_tri != REF_NONE) {
                const float4* q = reinterpret_cast<const float4*>(sc.isect + sc.last_tri);
                tri_test(q[0], q[1], q[2], sc.last_tri, r, h);
            }
        } else if (!still && slab(a, b, r, tn)) first = top_n ? root_ref : sc.root.ref;
        return first;
    }

    // next deferred sibling whose t_near does not exceed the best t (upstream tests this at pop time, raytrace.cu:55)
    __device__ __forceinline__ uint32_t pop(const Hit& h) {
        while (sp > 0) {
            --sp;
            uint2 e;
            if (sp < lds_slots) { e.x = stack[sp * stride].x; e.y = stack[sp * stride].y; }
            else e = spill[sp - lds_slots];
            if (!(__uint_as_float(e.y) > h.t)) return e.x;
        }
        return REF_NONE;
    }

    __device__ __forceinline__ bool at_inner() const { return !(cur & REF_LEAF); }        // REF_NONE has the leaf bit set
    __device__ __forceinline__ bool at_leaf() const { return cur != REF_NONE && (cur & REF_LEAF); }
    __device__ __forceinline__ bool done() const { return cur == REF_NONE; }

    __device__ __forceinline__ void push(uint32_t ref, float tn, TravCounters& tc) {
        const uint2 e = make_uint2(ref, __float_as_uint(tn));
        if (sp < lds_slots) { stack[sp * stride].x = e.x; stack[sp * stride].y = e.y; }
        else spill[sp - lds_slots] = e;
        sp++;
        if (COUNT) tc.max_sp = max(tc.max_sp, (uint32_t)sp);
    }

    // visit the two-level node `cur`: one 128-byte record decides up to four grandchildren, slot 0 = visited first
    __device__ __forceinline__ void node_step(const SceneDev& sc, const RayD& r, const Hit& h, TravCounters& tc) {
        f4v q0, q1, q2, q3, q4, q5, q6;             // the first 112 bytes of the record: 12 pairs of bounds, 4 references
        if (cur < n_hot) {
            const lds_float4* q = hot + cur * (RT_HOT_STRIDE / 16);
            q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3]; q4 = q[4]; q5 = q[5]; q6 = q[6];
        } else {
            const f4v* q = reinterpret_cast<const f4v*>((cur < top_n ? top : sc.nodes) + cur);
            q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3]; q4 = q[4]; q5 = q[5]; q6 = q[6];
        }
        const uint32_t r0 = __float_as_uint(q6.x), r1 = __float_as_uint(q6.y), r2 = __float_as_uint(q6.z), r3 = __float_as_uint(q6.w);
        if (COUNT) {
            tc.n_box += (r0 != REF_NONE) + (r1 != REF_NONE) + (r2 != REF_NONE) + (r3 != REF_NONE);
            tc.n_node++;
            tc.ray_visits++;
            tc.n_hot_visit += cur < n_hot;
            if (tc.hist && cur < RT_HIST_NODES) atomicAdd(tc.hist + cur, 1u);
            // one lane per wave counts the trip: sum over lanes = trips the wave executed
            if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) tc.w_node++;
        }
        // all four slots are tested without branches: an unused slot costs a few wasted operations, a branch per slot costs more
        float t0, t1, t2, t3;
        const bool h0 = (r0 != REF_NONE) & slab(q0.xy, q0.zw, q1.xy, r, t0) & !(t0 > h.t);
        const bool h1 = (r1 != REF_NONE) & slab(q1.zw, q2.xy, q2.zw, r, t1) & !(t1 > h.t);
        const bool h2 = (r2 != REF_NONE) & slab(q3.xy, q3.zw, q4.xy, r, t2) & !(t2 > h.t);
        const bool h3 = (r3 != REF_NONE) & slab(q4.zw, q5.xy, q5.zw, r, t3) & !(t3 > h.t);
        // The first hit slot is visited now; later hit slots wait on the stack, last slot deepest.  A slot whose
        // t_near already exceeds the best t is dropped now: the best t only shrinks (upstream would drop it at pop time).
        if (h3 && (h0 || h1 || h2)) push(r3, t3, tc);
        if (h2 && (h0 || h1)) push(r2, t2, tc);
        if (h1 && h0) push(r1, t1, tc);
        if (h0) cur = r0;
        else if (h1) cur = r1;
        else if (h2) cur = r2;
        else if (h3) cur = r3;
        else cur = pop(h);
    }

    // visit the leaf `cur`: left triangle, then right (raytrace.cu:57-62)
    __device__ __forceinline__ void leaf_step(const SceneDev& sc, const RayD& r, Hit& h, TravCounters& tc) {
        const uint32_t p = cur & REF_MASK;
        if (COUNT) {
            tc.n_tri += (cur & REF_PAIR) ? 2u : 1u;
            tc.ray_visits++;
            if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) tc.w_leaf++;
        }
        // both records of a two-triangle leaf are fetched at once (one memory round trip per leaf visit, not two); lanes at
        // a one-triangle leaf do not read a second record: every 16-byte piece a lane reads is an L1 access
        const f4v* q = reinterpret_cast<const f4v*>(sc.isect + p);
        const bool pair = (cur & REF_PAIR) != 0;
        const float4 a0 = as_float4(q[0]), a1 = as_float4(q[1]), a2 = as_float4(q[2]);
        // left unset without a second triangle: a default taken from the first record would make the second fetch wait for the first
        float4 b0, b1, b2;
        if (pair) { b0 = as_float4(q[3]); b1 = as_float4(q[4]); b2 = as_float4(q[5]); }
        tri_test(a0, a1, a2, p, r, h);
        if (pair) tri_test(b0, b1, b2, p + 1, r, h);
        cur = pop(h);
    }
};

template <bool COUNT>
__device__ __forceinline__ void closest_hit(const SceneDev& sc, const RayD& r, Hit& h, lds_uint2* stack, int stride, int lds_slots,
                                            TravCounters& tc) {
    uint2 spill[RT_SPILL_SLOTS];
    Walker<COUNT> w;
    w.init(stack, stride, lds_slots, spill);
    w.begin(sc, r, h, tc);
    for (;;) {
        while (w.at_inner()) w.node_step(sc, r, h, tc);
        if (w.done()) break;
        w.leaf_step(sc, r, h, tc);
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
// one iteration of upstream's depth loop after the closest-hit query (cuda/raytrace.cu:220-241):
// miss / light / Russian roulette / scatter.  Returns true when the path has ended (`color` is final),
// false when it goes on along `ray`.  `bounce` = scatters already done on this path (addresses the draws).
// ---------------------------------------------------------------------------
struct ShadeConst {
    v3 ambient;
    float rr;
    uint32_t k0, k1;
};

// What the samples of one semantic batch share at their first hit (they share the primary ray, cuda/raytrace.cu:210):
// everything triangle_interpolate computes (cuda/raytrace.cu:80-96) and the hit point.
struct HitPrep {
    bool hit;
    v3 rd, point, normal, hit_color;
    int mtype, textured;
    float optical_density, metal_fuzz;
};

__device__ __forceinline__ void prepare_hit(const SceneDev& sc, const RayD& ray, const Hit& h, HitPrep& q) {
    q.hit = h.tri != REF_NONE;
    q.textured = 0;
    q.mtype = RT_MAT_NONE;
    q.optical_density = 1.0f; q.metal_fuzz = 0.0f;
    q.rd = mk3(ray.dx, ray.dy, ray.dz);
    q.point = q.normal = q.hit_color = mk3(0.0f, 0.0f, 0.0f);
    if (!q.hit) return;
    // ---- triangle_interpolate, raytrace.cu:80-96 ----
    const float4* sq = reinterpret_cast<const float4*>(sc.shade + h.tri);
    const float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3];
    const float u = h.u, v = h.v;
    const float w = 1.0f - u - v;
    const v3 n0 = mk3(s0.x, s0.y, s0.z), n1 = mk3(s0.w, s1.x, s1.y), n2 = mk3(s1.z, s1.w, s2.x);
    q.normal = vnormalize(vadd(vadd(vscale(n0, w), vscale(n1, u)), vscale(n2, v)));
    const int mat_id = __float_as_int(s3.w);
    const float4* mq = reinterpret_cast<const float4*>(sc.mats + mat_id);
    const float4 m0 = mq[0], m1 = mq[1];
    q.mtype = __float_as_int(m0.x);
    q.optical_density = m1.x; q.metal_fuzz = m1.y;
    const int mtex = __float_as_int(m1.z);
    if (mtex < 0) {
        q.hit_color = mk3(m0.y, m0.z, m0.w);
    } else {
        const float tcx = w * s2.y + u * s2.w + v * s3.y;
        const float tcy = w * s2.z + u * s3.x + v * s3.z;
        q.textured = 1;
        q.hit_color = tex_fetch(sc, mtex, tcx, tcy);
    }
    q.point = vadd(mk3(ray.ox, ray.oy, ray.oz), vscale(q.rd, h.t));   // primitives.cuh:17-19
}

// The rest of the iteration for ONE sample: miss / light / roulette / scatter with that sample's draws.  `ray` is written
// only when the path goes on.  Counts are per sample (what upstream executes per sample), whoever shared the preparation.
// `repeatable` (set when the path goes on): the new ray does not depend on the sample's random vector -- a mirror, glass
// (one of two rays, by the sample's decision draw), a metal without fuzz on its reflecting branch -- so other samples
// that arrived along the same ray may leave along the same ray.
template <bool COUNT>
__device__ __forceinline__ bool scatter_hit(const HitPrep& q, const ShadeConst& k, RayD& ray, uint32_t pixel, uint32_t sample, uint32_t bounce,
                                            v3& color, uint32_t& n_shaded, uint32_t& n_tex, bool& repeatable) {
    repeatable = false;
    if (!q.hit) {
        color = vmul(color, k.ambient); return true;      // raytrace.cu:220-223
    } else {
        if (COUNT) { n_shaded++; n_tex += (uint32_t)q.textured; }
        const v3 normal = q.normal, hit_color = q.hit_color, rd = q.rd, point = q.point;
        const int mtype = q.mtype;
        const float optical_density = q.optical_density, metal_fuzz = q.metal_fuzz;
        uint32_t w_rr, w_dec, w2_, w3_;
        philox((uint32_t)pixel, (uint32_t)sample, bounce * 65536u, 0u, k.k0, k.k1, w_rr, w_dec, w2_, w3_);
        const float decision = u01(w_dec);
        if (mtype == RT_MAT_LIGHT) {
            color = vmul(color, hit_color); return true;                 // raytrace.cu:226-229
        } else if (u01(w_rr) < k.rr) {                                      // :230
            // ---- material_scatter, raytrace.cu:120-195 ----
            v3 dir = mk3(0.0f, 0.0f, 0.0f);
            bool scattered = true;
            const float dn = vdot(rd, normal);
            // both diffuse and metal surfaces start with the same draw; one loop serves the lanes of either kind
            // (two separate loops would each run as long as their slowest lane)
            v3 rnd = mk3(0.0f, 0.0f, 0.0f);
            if (mtype == RT_MAT_LAMBERTIAN || mtype == RT_MAT_METAL) rnd = unit_sphere((uint32_t)pixel, (uint32_t)sample, bounce, k.k0, k.k1);
            if (mtype == RT_MAT_LAMBERTIAN) {
                v3 d0 = normal;
                if (dn > 0.0f) d0 = vneg(normal);
                dir = vnormalize(vadd(d0, rnd));
                const float cos_theta = fabsf(vdot(dir, normal));
                color = vmul(color, vscale(hit_color, cos_theta));
            } else if (mtype == RT_MAT_METAL) {
                v3 nn = normal;
                if (dn > 0.0f) nn = vneg(normal);
                if (4.0f * decision < 3.0f + metal_fuzz) {
                    dir = vadd(nn, rnd);
                    const float cos_theta = fabsf(vdot(dir, nn));
                    color = vmul(color, vscale(hit_color, cos_theta));
                } else {
                    dir = vadd(vnormalize(reflect3(rd, nn)), vscale(rnd, metal_fuzz));
                    repeatable = metal_fuzz == 0.0f;
                }
            } else if (mtype == RT_MAT_REFLECTIVE) {
                dir = reflect3(rd, normal);
                color = vmul(color, hit_color);
                repeatable = true;
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
                bool refracted = false;
                {   // refract, raytrace.cu:102-110
                    const float dt = vdot(rd, outward);
                    const float disc = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
                    if (disc > 0.0f) {
                        refracted = true;
                        dir = vsub(vscale(vsub(rd, vscale(outward, dt)), ni_over_nt), vscale(outward, sqrtf(disc)));
                        float r0 = (1.0f - optical_density) / (1.0f + optical_density);   // schlick, :112-116
                        r0 = r0 * r0;
                        reflect_prob = r0 + (1.0f - r0) * powi5(1.0f - cosine);
                    }
                }
                // Without a refracted direction the ray reflects whatever the draw: upstream leaves the direction
                // uninitialised when the draw is exactly 1.0 (raytrace.cu:182-187), here that case reflects too.
                if (decision < reflect_prob || !refracted) dir = reflect3(rd, normal);
                color = vmul(color, hit_color);
                repeatable = true;
            } else {
                scattered = false;
            }
            if (scattered) {
                ray_set(ray, vadd(point, vscale(dir, 1e-4f)), dir);         // raytrace.cu:118
            } else {
                color = vmul(color, k.ambient); return true;             // :232-235
            }
        } else {
            color = vmul(color, k.ambient); return true;                 // :238-241
        }
    }
    return false;
}

template <bool COUNT>
__device__ __forceinline__ bool shade_hit(const SceneDev& sc, const ShadeConst& k, RayD& ray, const Hit& h, uint32_t pixel,
                                          uint32_t sample, uint32_t bounce, v3& color, uint32_t& n_shaded, uint32_t& n_tex) {
    HitPrep q;
    prepare_hit(sc, ray, h, q);
    bool repeatable;
    return scatter_hit<COUNT>(q, k, ray, pixel, sample, bounce, color, n_shaded, n_tex, repeatable);
}

template <bool COUNT>
__device__ __forceinline__ bool shade_hit(const SceneDev& sc, const ShadeConst& k, RayD& ray, const Hit& h, uint32_t pixel,
                                          uint32_t sample, uint32_t bounce, v3& color, uint32_t& n_shaded, uint32_t& n_tex, bool& repeatable) {
    HitPrep q;
    prepare_hit(sc, ray, h, q);
    return scatter_hit<COUNT>(q, k, ray, pixel, sample, bounce, color, n_shaded, n_tex, repeatable);
}

#endif
