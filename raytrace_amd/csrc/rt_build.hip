// rt_build.hip -- the scene tree built on the GPU (RT_OPT_GPU_BVH): one hierarchy over all triangles of all objects,
// emitted straight into the layout the kernels walk (Node4 / TriIsect / TriShade / TriOrigin, rt_packed.h).
//
// Upstream builds one median-split tree per object on the host, single-threaded (csrc/bvh.cpp:29-98), and the default
// upload keeps that tree node for node because the visiting order decides which of two equally near triangles wins and
// which boxes a ray is tested against.  This builder is the load-time alternative for large scenes: a different tree
// (so hit records are those of an exhaustive search up to the known disagreements between slab test and triangle test,
// not upstream's bit for bit), built in a fraction of a second where the host path takes seconds.
//
// Steps, all on the device:
//   1. per object the bounds of its triangles' centroids (ordered-integer atomics); sort key per triangle = object number,
//      then the Morton code of the centroid inside its object's bounds;
//   2. rocPRIM radix sort of (code, triangle);
//   3. leaves and hierarchy by agglomeration in Morton order (parallel locally-ordered clustering): every cluster looks a
//      few places to either side for the neighbour it would make the smallest box with, mutual choices are joined, the
//      array is compacted, until one cluster is left; two lone triangles that choose each other share a leaf (the walk's
//      leaves hold one or two triangles);
//   4. a node's box is the union of its children's, written when it is made;
//      (3' and 4', RT_BUILD_TREE=0 and scenes of one or two triangles: leaves = pairs of neighbours in sorted order, the
//      hierarchy the code bits give after Karras 2012, boxes bottom-up -- 13 to 40 % slower to walk)
//   5. the hierarchy is cut into four-slot records from the top (a record's slots: its root's children, then twice the
//      inner slot with the largest box opened); the records are numbered level by level, larger boxes first inside a level
//      (second radix sort), so that the records every walk starts with lead the array (the kernels keep a prefix in LDS);
//   6. stack need bottom-up over the two-level records; the slots of a record are ordered so that the slot with the
//      deepest subtree is visited last -- nothing waits on the stack while the walk is below it -- which keeps the
//      stack need logarithmic even where Morton codes make the tree deep;
//   7. triangle records leaf after leaf, the leaves in Morton order of their first triangle.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_build.h"

using namespace rtc;

namespace {

#define BUILD_BLOCK 256
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t LEAF_BIT = 0x80000000u;      // in the binary hierarchy: child is a leaf (low bits = leaf number)

struct RawTri { int32_t material_id; int32_t pad; float v[3][8]; };      // rt_triangle: position 0..2, normal 3..5, uv 6..7 per vertex
static_assert(sizeof(RawTri) == sizeof(rt_triangle), "RawTri");

__device__ __forceinline__ uint32_t ordered(float f) { const uint32_t b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__host__ __device__ __forceinline__ float unordered(uint32_t o) {
    const uint32_t b = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    float f;
    memcpy(&f, &b, 4);
    return f;
}

// the object whose range [first[k], first[k+1]) holds triangle g
__device__ __forceinline__ int object_of(const uint32_t* __restrict__ first, int n_objects, uint32_t g) {
    int lo = 0, hi = n_objects - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// bounds[6 * k + 0..2] = min, [6 * k + 3..5] = max of the centroids of object k, as ordered integers.  A block whose
// triangles all belong to one object (nearly all blocks) reduces in LDS and adds six atomics; a block that straddles
// objects lets every thread add its own.
__global__ __launch_bounds__(BUILD_BLOCK) void centroid_bounds_kernel(const RawTri* __restrict__ tris, uint32_t n, const uint32_t* __restrict__ first,
                                                                      int n_objects, uint32_t* bounds) {
    __shared__ uint32_t lo[3], hi[3];
    const uint32_t begin = blockIdx.x * BUILD_BLOCK, i = begin + threadIdx.x;
    const uint32_t end = min(n, begin + BUILD_BLOCK) - 1;
    const int obj_a = object_of(first, n_objects, begin), obj_b = object_of(first, n_objects, end);
    if (threadIdx.x < 3) { lo[threadIdx.x] = 0xFFFFFFFFu; hi[threadIdx.x] = 0u; }
    __syncthreads();
    if (i < n) {
        const RawTri& t = tris[i];
        const int mine = obj_a == obj_b ? obj_a : object_of(first, n_objects, i);
        for (int a = 0; a < 3; a++) {
            const float c = (t.v[0][a] + t.v[1][a] + t.v[2][a]) * (1.0f / 3.0f);
            if (!(c == c)) continue;
            if (obj_a == obj_b) { atomicMin(&lo[a], ordered(c)); atomicMax(&hi[a], ordered(c)); }
            else { atomicMin(&bounds[6 * mine + a], ordered(c)); atomicMax(&bounds[6 * mine + 3 + a], ordered(c)); }
        }
    }
    __syncthreads();
    if (obj_a == obj_b && threadIdx.x < 3) {
        atomicMin(&bounds[6 * obj_a + threadIdx.x], lo[threadIdx.x]);
        atomicMax(&bounds[6 * obj_a + 3 + threadIdx.x], hi[threadIdx.x]);
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t x) {      // 21 bits -> every third bit of 63
    uint64_t v = x & 0x1FFFFFu;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// Sort key of a triangle: its object's number in the high bits, below it the Morton code of its centroid inside the
// object's own centroid box (`bits` per axis).  The top of the hierarchy then divides the object sequence, and every
// object gets a subtree of its own: an object's box stays as tight as upstream's per-object trees make it, and a huge
// triangle (a ground plane) widens no box but those above its own object.
__global__ __launch_bounds__(BUILD_BLOCK) void morton_kernel(const RawTri* __restrict__ tris, uint32_t n, const uint32_t* __restrict__ first, int n_objects,
                                                             const uint32_t* __restrict__ bounds, int bits, uint64_t* __restrict__ keys,
                                                             uint32_t* __restrict__ ids) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const RawTri& t = tris[i];
    const int obj = object_of(first, n_objects, i);
    const uint32_t* ob = bounds + 6 * obj;
    // one cell size for all three axes (the largest extent sets it): cells are cubes, so the code's next bit halves
    // space the same way on every axis; per-axis scaling would cut a flat object into thin slabs
    float extent = 0.0f;
    for (int a = 0; a < 3; a++) extent = fmaxf(extent, unordered(ob[3 + a]) - unordered(ob[a]));
    const float cells = (float)(1u << bits);
    uint64_t key = 0;
    for (int a = 0; a < 3; a++) {
        const float lo = unordered(ob[a]);
        const float c = (t.v[0][a] + t.v[1][a] + t.v[2][a]) * (1.0f / 3.0f);
        float u = extent > 0.0f ? (c - lo) / extent : 0.0f;
        u = u == u ? fminf(fmaxf(u, 0.0f), 1.0f) : 0.0f;           // NaN centroids go to a corner
        const uint32_t q = (uint32_t)fminf(u * cells, cells - 1.0f);
        key |= spread21(q) << a;
    }
    keys[i] = ((uint64_t)obj << (3 * bits)) | key;
    ids[i] = i;
}

__global__ __launch_bounds__(BUILD_BLOCK) void leaf_keys_kernel(const uint64_t* __restrict__ sorted_keys, uint32_t m, uint64_t* __restrict__ leaf_keys) {
    const uint32_t j = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (j < m) leaf_keys[j] = sorted_keys[2 * (size_t)j];
}

// length of the common prefix of leaves i and j (equal codes: their positions decide); -1 outside the array
__device__ __forceinline__ int prefix(const uint64_t* __restrict__ keys, int m, int i, int j) {
    if (j < 0 || j >= m) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)(i ^ j));
    return __clzll((long long)(a ^ b));
}

struct Binary {               // the binary hierarchy over m leaves: m - 1 inner nodes, node 0 is the root
    uint32_t* child;          // [2 * (m-1)]: left, right; LEAF_BIT | leaf number, or inner node number
    uint32_t* parent;         // [m-1]: inner node above (NONE for the root)
    uint32_t* leaf_parent;    // [m]
    float* box;               // [6 * (m-1)]
    float* leaf_box;          // [6 * m]
    uint32_t* arrived;        // [m-1]
    uint32_t* depth;          // [m-1]
    // leaves made by the agglomeration itself (null otherwise: leaf j then holds the triangles at sorted places 2j, 2j+1):
    // a leaf's number is the sorted place of its first triangle, so the numbers are sparse in [0, n)
    uint32_t* leaf_size;      // [n] 0 = no such leaf, 1, 2
    uint32_t* leaf_other;     // [n] sorted place of the second triangle
    uint32_t* leaf_pos;       // [n+1] place of the leaf's first triangle in the packed arrays (running sum of the sizes)
};

__global__ __launch_bounds__(BUILD_BLOCK) void hierarchy_kernel(const uint64_t* __restrict__ keys, int m, Binary b) {
    const int i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= m - 1) return;
    const int d = prefix(keys, m, i, i + 1) - prefix(keys, m, i, i - 1) >= 0 ? 1 : -1;
    const int floor_prefix = prefix(keys, m, i, i - d);
    long long reach = 2;
    for (;;) {
        const long long at = (long long)i + reach * d;
        if (at < 0 || at >= m || !(prefix(keys, m, i, (int)at) > floor_prefix)) break;
        reach *= 2;
    }
    long long len = 0;
    for (long long t = reach / 2; t >= 1; t /= 2) {
        const long long at = (long long)i + (len + t) * d;
        if (at >= 0 && at < m && prefix(keys, m, i, (int)at) > floor_prefix) len += t;
    }
    const int j = (int)((long long)i + len * d);
    const int node_prefix = prefix(keys, m, i, j);
    long long s = 0, t = len;
    do {
        t = (t + 1) >> 1;
        const long long at = (long long)i + (s + t) * d;
        if (at >= 0 && at < m && prefix(keys, m, i, (int)at) > node_prefix) s += t;
    } while (t > 1);
    const int split = (int)((long long)i + s * d) + (d < 0 ? -1 : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const uint32_t left = first == split ? (LEAF_BIT | (uint32_t)split) : (uint32_t)split;
    const uint32_t right = last == split + 1 ? (LEAF_BIT | (uint32_t)(split + 1)) : (uint32_t)(split + 1);
    b.child[2 * (size_t)i] = left;
    b.child[2 * (size_t)i + 1] = right;
    if (left & LEAF_BIT) b.leaf_parent[split] = (uint32_t)i; else b.parent[split] = (uint32_t)i;
    if (right & LEAF_BIT) b.leaf_parent[split + 1] = (uint32_t)i; else b.parent[split + 1] = (uint32_t)i;
    if (i == 0) b.parent[0] = NONE;
}

// Values another workgroup wrote earlier in the same launch are read and written at device scope, past the caches that
// are private to a CU or an XCD.
__device__ __forceinline__ float peek(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void poke(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t peek(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void poke(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A leaf's box is the exact min/max of its triangles' corners, as upstream's (csrc/bvh.cpp:36-47).
__global__ __launch_bounds__(BUILD_BLOCK) void fit_kernel(const RawTri* __restrict__ tris, const uint32_t* __restrict__ sorted_ids, uint32_t n, uint32_t m, Binary b) {
    const uint32_t j = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (j >= m) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t k = 2 * j; k < 2 * j + 2 && k < n; k++) {
        const RawTri& t = tris[sorted_ids[k]];
        for (int c = 0; c < 3; c++)
            for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], t.v[c][a]); hi[a] = fmaxf(hi[a], t.v[c][a]); }
    }
    for (int a = 0; a < 3; a++) { poke(&b.leaf_box[6 * (size_t)j + a], lo[a]); poke(&b.leaf_box[6 * (size_t)j + 3 + a], hi[a]); }
    if (m == 1) return;
    uint32_t node = b.leaf_parent[j];
    while (node != NONE) {
        __threadfence();
        if (atomicAdd(&b.arrived[node], 1u) == 0u) return;          // the first to arrive leaves the node to the second
        __threadfence();
        for (int side = 0; side < 2; side++) {
            const uint32_t c = b.child[2 * (size_t)node + side];
            const float* src = (c & LEAF_BIT) ? b.leaf_box + 6 * (size_t)(c & ~LEAF_BIT) : b.box + 6 * (size_t)c;
            for (int a = 0; a < 3; a++) {
                const float l = peek(src + a), h = peek(src + 3 + a);
                lo[a] = side == 0 ? l : fminf(lo[a], l);
                hi[a] = side == 0 ? h : fmaxf(hi[a], h);
            }
        }
        for (int a = 0; a < 3; a++) { poke(&b.box[6 * (size_t)node + a], lo[a]); poke(&b.box[6 * (size_t)node + 3 + a], hi[a]); }
        node = b.parent[node];
    }
}

// ---------------------------------------------------------------------------
// The hierarchy by agglomeration (parallel locally-ordered clustering, Meister & Bittner 2018) instead of by the bits
// of the codes: the clusters, kept in Morton order, each look RADIUS places to either side for the neighbour with
// which they would make the smallest box; pairs that chose each other are joined; the array is compacted; repeat
// until one cluster is left (the pair that makes the smallest box of a round always chose each other, so every round
// joins something; a round joins 30-45 % of the clusters).  The Morton order only decides who is near enough to be
// asked, the boxes decide who is joined, which is what the code-bit hierarchy lacks.  An object's clusters join among
// themselves until the object is one cluster -- its subtree then has the object's own tight box, and a ground plane next
// to a car part in the order does not swallow it -- and whole objects join each other the same way.
// Measured against the code-bit hierarchy over blind pairs: formula1 6422 -> 9083 Mrays/s, 30 M triangles 2155 -> 3621
// (upstream's median-split trees: 8162 and 3236), and the build is quicker (30 M triangles: 170 -> 92 ms).  A wider
// search (12 to 32 places) changes little.
// ---------------------------------------------------------------------------
#define PLOC_RADIUS 8
#define PLOC_OBJECT_RADIUS 1024
constexpr uint32_t WHOLE_BIT = 0x80000000u;     // object word: the cluster holds every leaf of its object (or of several objects)

struct Clusters {
    float4* lo;            // box min; w = object word: object number | WHOLE_BIT
    float4* hi;            // box max; w = what the cluster is in the hierarchy: LEAF_BIT | leaf number, or inner node number
    uint32_t* leaves;      // leaves below it
};

// `objects_join`: 0 while parts of objects are left -- whole objects then wait, because their places in the array follow
// the object numbers, which say nothing about where they are; once only whole objects are left they are few, the
// search is as wide as the array (up to PLOC_OBJECT_RADIUS), and the tree above the objects is a true agglomeration.
__global__ __launch_bounds__(BUILD_BLOCK) void nearest_kernel(uint32_t count, Clusters c, uint32_t radius, int objects_join, uint32_t* __restrict__ nearest,
                                                              uint32_t* parts_left) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= count) return;
    const float4 lo = c.lo[i], hi = c.hi[i];
    const uint32_t mine = __float_as_uint(lo.w);
    uint32_t best = NONE;
    float best_area = INFINITY;
    if (!(mine & WHOLE_BIT)) *parts_left = 1u;            // every writer writes the same value
    else if (!objects_join) { nearest[i] = NONE; return; }
    const uint32_t from = i >= radius ? i - radius : 0u, to = min(count - 1u, i + radius);
    for (uint32_t j = from; j <= to; j++) {
        if (j == i) continue;
        const float4 l2 = c.lo[j], h2 = c.hi[j];
        const uint32_t theirs = __float_as_uint(l2.w);
        // parts of one object among themselves, whole objects among themselves
        if (!(mine == theirs && !(mine & WHOLE_BIT)) && !((mine & WHOLE_BIT) && (theirs & WHOLE_BIT))) continue;
        const float dx = fmaxf(hi.x, h2.x) - fminf(lo.x, l2.x), dy = fmaxf(hi.y, h2.y) - fminf(lo.y, l2.y), dz = fmaxf(hi.z, h2.z) - fminf(lo.z, l2.z);
        float area = dx * dy + dy * dz + dz * dx;
        if (!(area == area)) area = INFINITY;
        if (best == NONE || area < best_area) { best = j; best_area = area; }      // equal areas: the lower place
    }
    nearest[i] = best;
}

__global__ __launch_bounds__(BUILD_BLOCK) void compact_kernel(uint32_t count, Clusters from, Clusters to, const uint32_t* __restrict__ goes_on, const uint32_t* __restrict__ place) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= count || !goes_on[i]) return;
    const uint32_t at = place[i];
    to.lo[at] = from.lo[i]; to.hi[at] = from.hi[i]; to.leaves[at] = from.leaves[i];
}

// It starts from the single triangles, so that it also decides which two triangles share a leaf: two lone triangles
// that choose each other become a two-triangle leaf, a lone triangle that joins anything else becomes a leaf of its
// own.  (Agglomeration over blind pairs of sorted neighbours cost 75 % more triangle tests per ray than upstream's
// leaves: 7.6 against 4.3; leaves chosen this way: 4.0.)
constexpr uint32_t TRI_BIT = 0x40000000u;       // cluster reference: a lone triangle, low bits = its sorted place

__global__ __launch_bounds__(BUILD_BLOCK) void triangle_clusters_kernel(const RawTri* __restrict__ tris, const uint32_t* __restrict__ sorted_ids, uint32_t n,
                                                                        const uint32_t* __restrict__ first, int n_objects, Clusters c) {
    const uint32_t k = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (k >= n) return;
    const uint32_t g = sorted_ids[k];
    const RawTri& t = tris[g];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int v = 0; v < 3; v++)
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], t.v[v][a]); hi[a] = fmaxf(hi[a], t.v[v][a]); }
    const uint32_t obj = (uint32_t)object_of(first, n_objects, g);
    const uint32_t whole = first[obj + 1] - first[obj] == 1u ? WHOLE_BIT : 0u;
    c.lo[k] = make_float4(lo[0], lo[1], lo[2], __uint_as_float(obj | whole));
    c.hi[k] = make_float4(hi[0], hi[1], hi[2], __uint_as_float(TRI_BIT | k));
    c.leaves[k] = 1u;                       // here: triangles below the cluster
}

__global__ __launch_bounds__(BUILD_BLOCK) void merge_triangles_kernel(uint32_t count, Clusters c, Clusters next, const uint32_t* __restrict__ nearest, Binary b,
                                                                      uint32_t* made, const uint32_t* __restrict__ first, uint32_t* __restrict__ goes_on) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t partner = nearest[i];
    const bool joined = partner != NONE && nearest[partner] == i;
    if (joined && i > partner) { goes_on[i] = 0u; return; }
    float4 lo = c.lo[i], hi = c.hi[i];
    uint32_t below = c.leaves[i];
    if (joined) {
        const float4 l2 = c.lo[partner], h2 = c.hi[partner];
        uint32_t kids[2] = {__float_as_uint(hi.w), __float_as_uint(h2.w)};
        const float4 klo[2] = {lo, l2}, khi[2] = {hi, h2};
        auto leaf_of = [&](uint32_t place, uint32_t second, const float4& bl, const float4& bh, uint32_t size) {
            b.leaf_size[place] = size;
            b.leaf_other[place] = second;
            float* box = b.leaf_box + 6 * (size_t)place;
            box[0] = bl.x; box[1] = bl.y; box[2] = bl.z; box[3] = bh.x; box[4] = bh.y; box[5] = bh.z;
            return LEAF_BIT | place;
        };
        const uint32_t o1 = __float_as_uint(lo.w), o2 = __float_as_uint(l2.w);
        below += c.leaves[partner];
        uint32_t word;
        if (o1 == o2 && !(o1 & WHOLE_BIT)) word = o1 | (below == first[o1 + 1] - first[o1] ? WHOLE_BIT : 0u);
        else word = min(o1 & ~WHOLE_BIT, o2 & ~WHOLE_BIT) | WHOLE_BIT;
        const float4 ulo = make_float4(fminf(lo.x, l2.x), fminf(lo.y, l2.y), fminf(lo.z, l2.z), __uint_as_float(word));
        float4 uhi = make_float4(fmaxf(hi.x, h2.x), fmaxf(hi.y, h2.y), fmaxf(hi.z, h2.z), 0.0f);
        uint32_t made_ref;
        if ((kids[0] & TRI_BIT) && (kids[1] & TRI_BIT) && !(kids[0] & LEAF_BIT) && !(kids[1] & LEAF_BIT)) {
            // two lone triangles: one leaf, numbered by the earlier place (the array stays in order of first places)
            made_ref = leaf_of(kids[0] & ~TRI_BIT, kids[1] & ~TRI_BIT, ulo, uhi, 2u);
        } else {
            for (int side = 0; side < 2; side++)
                if ((kids[side] & TRI_BIT) && !(kids[side] & LEAF_BIT)) kids[side] = leaf_of(kids[side] & ~TRI_BIT, NONE, klo[side], khi[side], 1u);
            const uint32_t node = atomicAdd(made, 1u);
            for (int side = 0; side < 2; side++) {
                b.child[2 * (size_t)node + side] = kids[side];
                if (kids[side] & LEAF_BIT) b.leaf_parent[kids[side] & ~LEAF_BIT] = node; else b.parent[kids[side]] = node;
            }
            b.parent[node] = NONE;              // until something joins it: the last node made stays the root
            float* box = b.box + 6 * (size_t)node;
            box[0] = ulo.x; box[1] = ulo.y; box[2] = ulo.z; box[3] = uhi.x; box[4] = uhi.y; box[5] = uhi.z;
            made_ref = node;
        }
        lo = ulo;
        uhi.w = __uint_as_float(made_ref);
        hi = uhi;
    }
    next.lo[i] = lo; next.hi[i] = hi; next.leaves[i] = below;
    goes_on[i] = 1u;
}

// the packed place of every triangle: leaves in the order of their numbers, a leaf's second triangle behind its first
__global__ __launch_bounds__(BUILD_BLOCK) void packed_order_kernel(uint32_t n, Binary b, const uint32_t* __restrict__ sorted_ids, uint32_t* __restrict__ packed_ids) {
    const uint32_t k = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (k >= n || b.leaf_size[k] == 0u) return;
    const uint32_t at = b.leaf_pos[k];
    packed_ids[at] = sorted_ids[k];
    if (b.leaf_size[k] == 2u) packed_ids[at + 1] = sorted_ids[b.leaf_other[k]];
}

// depth of the binary hierarchy (a statistic; the records are cut by mark_records_kernel)
__global__ __launch_bounds__(BUILD_BLOCK) void depth_kernel(uint32_t inner, Binary b, uint32_t* max_depth) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= inner) return;
    uint32_t d = 0;
    for (uint32_t p = b.parent[i]; p != NONE; p = b.parent[p]) d++;
    b.depth[i] = d;
    atomicMax(max_depth, d);
}

__device__ __forceinline__ float half_area(const float* bx) {
    const float dx = bx[3] - bx[0], dy = bx[4] - bx[1], dz = bx[5] - bx[2];
    const float a = dx * dy + dy * dz + dz * dx;
    return (a == a && a > 0.0f) ? a : 0.0f;
}

// The up to four slots of the record rooted at binary node `node`: its two children, and then, twice, the inner slot
// with the largest box replaced by its own two children in place.  A ray is likelier to enter a larger box, so the
// boxes most worth deciding in this visit are the ones below it; on a balanced tree this is the children's children,
// on the lopsided trees an agglomeration makes it follows the large side down.
__device__ __forceinline__ uint32_t record_slots(const Binary& b, uint32_t node, uint32_t slots[4]) {
    uint32_t used = 2;
    slots[0] = b.child[2 * (size_t)node];
    slots[1] = b.child[2 * (size_t)node + 1];
    for (int turn = 0; turn < 2; turn++) {
        int widest = -1;
        float area = -1.0f;
        for (uint32_t k = 0; k < used; k++) {
            if (slots[k] & LEAF_BIT) continue;
            const float a = half_area(b.box + 6 * (size_t)slots[k]);
            if (a > area) { area = a; widest = (int)k; }
        }
        if (widest < 0) break;
        const uint32_t c = slots[widest];
        for (uint32_t k = used; k > (uint32_t)widest + 1; k--) slots[k] = slots[k - 1];
        slots[widest] = b.child[2 * (size_t)c];
        slots[widest + 1] = b.child[2 * (size_t)c + 1];
        used++;
    }
    return used;
}

// One level of records: every root of the frontier is marked and numbered with its level, the inner slots of its record
// are the next frontier.
__global__ __launch_bounds__(BUILD_BLOCK) void mark_records_kernel(const uint32_t* __restrict__ frontier, uint32_t count, Binary b, uint32_t level,
                                                                   uint32_t* __restrict__ is_root, uint32_t* __restrict__ level_of,
                                                                   uint32_t* __restrict__ next, uint32_t* next_count) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= count) return;
    const uint32_t node = frontier[i];
    is_root[node] = 1u;
    level_of[node] = level;
    uint32_t slots[4];
    const uint32_t used = record_slots(b, node, slots);
    for (uint32_t k = 0; k < used; k++)
        if (!(slots[k] & LEAF_BIT)) next[atomicAdd(next_count, 1u)] = slots[k];
}

// the roots of the records, compacted: their sort key (level, then falling box area) and their number
__global__ __launch_bounds__(BUILD_BLOCK) void even_keys_kernel(uint32_t inner, Binary b, const uint32_t* __restrict__ even_flag, const uint32_t* __restrict__ even_pos,
                                                                const uint32_t* __restrict__ level_of, uint64_t* __restrict__ keys, uint32_t* __restrict__ nodes) {
    const uint32_t i = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (i >= inner || !even_flag[i]) return;
    const uint32_t at = even_pos[i];
    keys[at] = ((uint64_t)min(level_of[i], 0xFFFFu) << 32) | (uint64_t)(0xFFFFFFFFu - __float_as_uint(half_area(b.box + 6 * (size_t)i)));
    nodes[at] = i;
}

__global__ __launch_bounds__(BUILD_BLOCK) void rank_kernel(uint32_t n4, const uint32_t* __restrict__ sorted_nodes, uint32_t* __restrict__ rank_of) {
    const uint32_t r = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (r < n4) rank_of[sorted_nodes[r]] = r;
}

struct Records {
    Node4* nodes4;
    uint32_t* parent4;        // [n4]
    uint32_t* inner_slots;    // [n4] slots that lead to another record
    uint32_t* arrived4;       // [n4]
    uint32_t* need4;          // [n4] stack entries a walk below the record can have waiting at once
};

__device__ __forceinline__ uint32_t leaf_ref(const Binary& b, uint32_t leaf, uint32_t n) {
    if (b.leaf_pos) return REF_LEAF | b.leaf_pos[leaf] | (b.leaf_size[leaf] == 2u ? REF_PAIR : 0u);
    const uint32_t first = 2 * leaf;
    return REF_LEAF | first | (first + 1 < n ? REF_PAIR : 0u);
}

// One thread per record (in its final place): the slots record_slots chooses below its root.
__global__ __launch_bounds__(BUILD_BLOCK) void emit_records_kernel(uint32_t n4, uint32_t n, const uint32_t* __restrict__ sorted_nodes,
                                                                   const uint32_t* __restrict__ rank_of, Binary b, Records rec) {
    const uint32_t r = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (r >= n4) return;
    const uint32_t node = sorted_nodes[r];
    Node4 out;
    uint32_t used = 0, inner = 0;
    auto put = [&](uint32_t c) {
        const float* bx = (c & LEAF_BIT) ? b.leaf_box + 6 * (size_t)(c & ~LEAF_BIT) : b.box + 6 * (size_t)c;
        for (int a = 0; a < 6; a++) out.box[used][a] = bx[a];
        if (c & LEAF_BIT) {
            out.ref[used] = leaf_ref(b, c & ~LEAF_BIT, n);
        } else {
            const uint32_t rr = rank_of[c];
            out.ref[used] = rr;
            rec.parent4[rr] = r;
            inner++;
        }
        used++;
    };
    uint32_t slots[4];
    const uint32_t chosen = record_slots(b, node, slots);
    for (uint32_t k = 0; k < chosen; k++) put(slots[k]);
    for (uint32_t k = used; k < 4; k++) {
        for (int a = 0; a < 3; a++) { out.box[k][a] = INFINITY; out.box[k][3 + a] = -INFINITY; }
        out.ref[k] = REF_NONE;
    }
    for (int k = 0; k < 4; k++) out.pad[k] = 0;
    rec.nodes4[r] = out;
    rec.inner_slots[r] = inner;
    if (r == 0) rec.parent4[0] = NONE;
}

// Stack need, bottom-up.  While the walk is below slot i of a record, the later slots may all wait on the stack: need =
// max over slots of (slots behind it + need below it).  The slots are re-ordered by rising need, so the deepest subtree
// comes last and adds nothing of its own record.  A record is finished by whichever thread brings the last of its inner slots.
__global__ __launch_bounds__(BUILD_BLOCK) void need_kernel(uint32_t n4, Records rec, int reorder) {
    const uint32_t start = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (start >= n4 || rec.inner_slots[start] != 0) return;
    uint32_t r = start;
    for (;;) {
        Node4 nd = rec.nodes4[r];
        uint32_t need[4], used = 0;
        for (int k = 0; k < 4; k++) {
            if (nd.ref[k] == REF_NONE) continue;
            need[k] = (nd.ref[k] & REF_LEAF) ? 0u : peek(&rec.need4[nd.ref[k]]);
            used++;
        }
        // insertion sort of the used slots (they are the first `used`) by need, stable
        for (uint32_t a = 1; reorder && a < used; a++)
            for (uint32_t k = a; k > 0 && need[k - 1] > need[k]; k--) {
                const uint32_t tn = need[k]; need[k] = need[k - 1]; need[k - 1] = tn;
                const uint32_t tr = nd.ref[k]; nd.ref[k] = nd.ref[k - 1]; nd.ref[k - 1] = tr;
                for (int c = 0; c < 6; c++) { const float tb = nd.box[k][c]; nd.box[k][c] = nd.box[k - 1][c]; nd.box[k - 1][c] = tb; }
            }
        uint32_t total = 0;
        for (uint32_t k = 0; k < used; k++) total = max(total, used - 1 - k + need[k]);
        rec.nodes4[r] = nd;
        poke(&rec.need4[r], total);
        const uint32_t up = rec.parent4[r];
        if (up == NONE) return;
        __threadfence();
        if (atomicAdd(&rec.arrived4[up], 1u) + 1u != rec.inner_slots[up]) return;
        __threadfence();
        r = up;
    }
}

__global__ __launch_bounds__(BUILD_BLOCK) void emit_triangles_kernel(const RawTri* __restrict__ tris, const uint32_t* __restrict__ sorted_ids, uint32_t n,
                                                                     const uint32_t* __restrict__ object_first, int n_objects,
                                                                     TriIsect* __restrict__ isect, TriShade* __restrict__ shade, TriOrigin* __restrict__ origin) {
    const uint32_t p = blockIdx.x * BUILD_BLOCK + threadIdx.x;
    if (p >= n) return;
    const uint32_t g = sorted_ids[p];
    const RawTri& t = tris[g];
    TriIsect r;
    for (int a = 0; a < 3; a++) {
        r.v0[a] = t.v[0][a];
        r.e1[a] = t.v[1][a] + (-t.v[0][a]);           // the two edge subtractions of cuda/raytrace.cu:25-26, done once
        r.e2[a] = t.v[2][a] + (-t.v[0][a]);
        r.pad[a] = 0;
    }
    isect[p] = r;
    TriShade s;
    for (int a = 0; a < 3; a++) { s.n0[a] = t.v[0][3 + a]; s.n1[a] = t.v[1][3 + a]; s.n2[a] = t.v[2][3 + a]; }
    for (int a = 0; a < 2; a++) { s.uv0[a] = t.v[0][6 + a]; s.uv1[a] = t.v[1][6 + a]; s.uv2[a] = t.v[2][6 + a]; }
    s.material_id = t.material_id;
    shade[p] = s;
    const int obj = object_of(object_first, n_objects, g);
    origin[p] = TriOrigin{obj, (int32_t)(g - object_first[obj])};
}

// The triangle a walk that enters every box tests last: the last used slot all the way down, then the leaf's last triangle.
__global__ void last_triangle_kernel(const Node4* __restrict__ nodes4, uint32_t root_ref, uint32_t* out) {
    uint32_t ref = root_ref;
    for (int guard = 0; guard < 4096 && ref != REF_NONE && !(ref & REF_LEAF); guard++) {
        uint32_t last = REF_NONE;
        for (int k = 0; k < 4; k++) if (nodes4[ref].ref[k] != REF_NONE) last = nodes4[ref].ref[k];
        ref = last;
    }
    *out = (ref == REF_NONE || !(ref & REF_LEAF)) ? REF_NONE : (ref & REF_MASK) + ((ref & REF_PAIR) ? 1u : 0u);
}

struct Scratch {              // device allocations that live only during the build
    std::vector<void*> held;
    ~Scratch() { for (void* p : held) (void)hipFree(p); }
    template <class T> hipError_t get(T** p, size_t count) {
        const hipError_t e = hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) held.push_back(*p);
        return e;
    }
};

unsigned blocks_for(size_t n) { return (unsigned)((n + BUILD_BLOCK - 1) / BUILD_BLOCK); }

}  // namespace

namespace rtc {

#define BUILD_TRY(expr)                                                            \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); release_device_scene(out); return -1; } \
    } while (0)

void release_device_scene(DeviceScene& s) {
    void** ptrs[] = {&s.nodes4, &s.isect, &s.shade, &s.origin};
    for (void** p : ptrs) { if (*p) { (void)hipFree(*p); *p = nullptr; } }
}

// -2: the agglomeration made too little progress (it joins at least one pair a round, but a scene whose neighbours'
// distances shrink steadily along the order joins ONLY one): the caller builds the code-bit hierarchy instead
#define PLOC_MAX_ROUNDS 256
static int build_impl(const rt_object* objects, int n_objects, DeviceScene& out, std::string& err, bool may_agglomerate) {
    out = DeviceScene();
    uint64_t total = 0;
    std::vector<uint32_t> first(n_objects + 1, 0);
    for (int i = 0; i < n_objects; i++) {
        if (!objects[i].triangles || objects[i].n_triangles <= 0) { err = "object " + std::to_string(i) + ": no triangles"; return -1; }
        first[i] = (uint32_t)total;
        total += (uint64_t)objects[i].n_triangles;
        if (total > (uint64_t)REF_MASK) { err = "scene exceeds 2^30-1 triangles"; return -1; }
    }
    first[n_objects] = (uint32_t)total;
    const uint32_t n = (uint32_t)total;
    // which hierarchy: by agglomeration from the single triangles, which also chooses the leaves (default), or -- for one
    // or two triangles, and as the experiment switch RT_BUILD_TREE=0 -- the one the code bits give over pairs of sorted
    // neighbours (Karras 2012)
    const char* tree_env = std::getenv("RT_BUILD_TREE");
    const bool agglomerate = may_agglomerate && n >= 3 && !(tree_env && std::atoi(tree_env) == 0);
    const uint32_t m = agglomerate ? n : (n + 1) / 2;          // leaves (agglomeration: room for as many, numbered sparsely)
    uint32_t inner = m - 1;                                     // inner nodes (agglomeration: as many as turn out)
    Scratch tmp;
    hipEvent_t ev[3];
    for (hipEvent_t& e : ev) BUILD_TRY(hipEventCreate(&e));
    struct EventGuard { hipEvent_t* ev; ~EventGuard() { for (int k = 0; k < 3; k++) (void)hipEventDestroy(ev[k]); } } guard{ev};

    // ---- the triangles as the caller holds them, object after object ----
    RawTri* d_tris = nullptr;
    uint32_t* d_first = nullptr;
    BUILD_TRY(tmp.get(&d_tris, n));
    BUILD_TRY(tmp.get(&d_first, (size_t)n_objects + 1));
    BUILD_TRY(hipEventRecord(ev[0], nullptr));
    for (int i = 0; i < n_objects; i++)
        BUILD_TRY(hipMemcpyAsync(d_tris + first[i], objects[i].triangles, (size_t)objects[i].n_triangles * sizeof(RawTri), hipMemcpyHostToDevice, nullptr));
    BUILD_TRY(hipMemcpyAsync(d_first, first.data(), first.size() * sizeof(uint32_t), hipMemcpyHostToDevice, nullptr));
    BUILD_TRY(hipEventRecord(ev[1], nullptr));

    // ---- outputs ----
    BUILD_TRY(hipMalloc(&out.isect, ((size_t)n + 1) * sizeof(TriIsect)));      // one spare record behind the last triangle
    BUILD_TRY(hipMalloc(&out.shade, (size_t)n * sizeof(TriShade)));
    BUILD_TRY(hipMalloc(&out.origin, (size_t)n * sizeof(TriOrigin)));
    BUILD_TRY(hipMemsetAsync((TriIsect*)out.isect + n, 0, sizeof(TriIsect), nullptr));

    // ---- 1, 2: Morton codes, sorted ----
    uint32_t* d_bounds = nullptr;
    uint64_t *d_keys = nullptr, *d_keys_sorted = nullptr;
    uint32_t *d_ids = nullptr, *d_ids_sorted = nullptr;
    BUILD_TRY(tmp.get(&d_bounds, 6 * (size_t)n_objects));
    BUILD_TRY(tmp.get(&d_keys, n));
    BUILD_TRY(tmp.get(&d_keys_sorted, n));
    BUILD_TRY(tmp.get(&d_ids, n));
    BUILD_TRY(tmp.get(&d_ids_sorted, n));
    std::vector<uint32_t> bounds_init(6 * (size_t)n_objects);
    for (int i = 0; i < n_objects; i++)
        for (int a = 0; a < 3; a++) { bounds_init[6 * i + a] = 0xFFFFFFFFu; bounds_init[6 * i + 3 + a] = 0u; }
    BUILD_TRY(hipMemcpy(d_bounds, bounds_init.data(), bounds_init.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    int object_bits = 0;
    while ((1ll << object_bits) < n_objects) object_bits++;
    const int axis_bits = std::min(21, (63 - object_bits) / 3);
    hipLaunchKernelGGL(centroid_bounds_kernel, dim3(blocks_for(n)), dim3(BUILD_BLOCK), 0, nullptr, d_tris, n, d_first, n_objects, d_bounds);
    hipLaunchKernelGGL(morton_kernel, dim3(blocks_for(n)), dim3(BUILD_BLOCK), 0, nullptr, d_tris, n, d_first, n_objects, d_bounds, axis_bits, d_keys, d_ids);
    BUILD_TRY(hipGetLastError());
    {
        size_t bytes = 0;
        BUILD_TRY(rocprim::radix_sort_pairs(nullptr, bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, (size_t)n, 0, 63, nullptr));
        char* d_temp = nullptr;
        BUILD_TRY(tmp.get(&d_temp, bytes));
        BUILD_TRY(rocprim::radix_sort_pairs(d_temp, bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, (size_t)n, 0, 63, nullptr));
    }
    // ---- 3, 4: leaves, hierarchy, boxes ----
    Binary b = {};
    uint64_t* d_leaf_keys = nullptr;
    BUILD_TRY(tmp.get(&d_leaf_keys, m));
    BUILD_TRY(tmp.get(&b.child, 2 * (size_t)inner));
    BUILD_TRY(tmp.get(&b.parent, inner));
    BUILD_TRY(tmp.get(&b.leaf_parent, m));
    BUILD_TRY(tmp.get(&b.box, 6 * (size_t)inner));
    BUILD_TRY(tmp.get(&b.leaf_box, 6 * (size_t)m));
    BUILD_TRY(tmp.get(&b.arrived, inner));
    BUILD_TRY(tmp.get(&b.depth, inner));
    BUILD_TRY(hipMemsetAsync(b.arrived, 0, std::max<size_t>(inner, 1) * sizeof(uint32_t), nullptr));
    const uint32_t* d_packed_ids = d_ids_sorted;               // the triangle at every place of the packed arrays
    uint32_t root_node = 0;
    if (!agglomerate) {
        hipLaunchKernelGGL(leaf_keys_kernel, dim3(blocks_for(m)), dim3(BUILD_BLOCK), 0, nullptr, d_keys_sorted, m, d_leaf_keys);
        if (inner) hipLaunchKernelGGL(hierarchy_kernel, dim3(blocks_for(inner)), dim3(BUILD_BLOCK), 0, nullptr, d_leaf_keys, (int)m, b);
        hipLaunchKernelGGL(fit_kernel, dim3(blocks_for(m)), dim3(BUILD_BLOCK), 0, nullptr, d_tris, d_ids_sorted, n, m, b);
        BUILD_TRY(hipGetLastError());
    } else {
        Clusters cl[2] = {};
        uint32_t *d_nearest = nullptr, *d_goes_on = nullptr, *d_place = nullptr, *d_made = nullptr, *d_parts_left = nullptr;
        bool objects_join = false;
        for (Clusters& c : cl) {
            BUILD_TRY(tmp.get(&c.lo, m));
            BUILD_TRY(tmp.get(&c.hi, m));
            BUILD_TRY(tmp.get(&c.leaves, m));
        }
        BUILD_TRY(tmp.get(&d_nearest, m));
        BUILD_TRY(tmp.get(&b.leaf_size, n));
        BUILD_TRY(tmp.get(&b.leaf_other, n));
        BUILD_TRY(tmp.get(&b.leaf_pos, (size_t)n + 1));
        BUILD_TRY(hipMemsetAsync(b.leaf_size, 0, (size_t)n * sizeof(uint32_t), nullptr));
        BUILD_TRY(tmp.get(&d_goes_on, (size_t)m + 1));
        BUILD_TRY(tmp.get(&d_place, (size_t)m + 1));
        BUILD_TRY(tmp.get(&d_made, 1));
        BUILD_TRY(tmp.get(&d_parts_left, 1));
        BUILD_TRY(hipMemsetAsync(d_made, 0, sizeof(uint32_t), nullptr));
        hipLaunchKernelGGL(triangle_clusters_kernel, dim3(blocks_for(n)), dim3(BUILD_BLOCK), 0, nullptr, d_tris, d_ids_sorted, n, d_first, n_objects, cl[0]);
        BUILD_TRY(hipGetLastError());
        size_t scan_bytes = 0;
        BUILD_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_goes_on, d_place, 0u, (size_t)m + 1, rocprim::plus<uint32_t>(), nullptr));
        char* d_scan_temp = nullptr;
        BUILD_TRY(tmp.get(&d_scan_temp, scan_bytes));
        const char* radius_env = std::getenv("RT_BUILD_RADIUS");
        const uint32_t radius = radius_env ? (uint32_t)std::max(1, std::atoi(radius_env)) : PLOC_RADIUS;
        uint32_t count = m;
        for (int round = 0; count > 1; round++) {
            BUILD_TRY(hipMemsetAsync(d_parts_left, 0, sizeof(uint32_t), nullptr));
            hipLaunchKernelGGL(nearest_kernel, dim3(blocks_for(count)), dim3(BUILD_BLOCK), 0, nullptr, count, cl[0], objects_join ? (uint32_t)PLOC_OBJECT_RADIUS : radius,
                               objects_join ? 1 : 0, d_nearest, d_parts_left);
            hipLaunchKernelGGL(merge_triangles_kernel, dim3(blocks_for(count)), dim3(BUILD_BLOCK), 0, nullptr, count, cl[0], cl[1], d_nearest, b, d_made, d_first, d_goes_on);
            BUILD_TRY(hipGetLastError());
            BUILD_TRY(hipMemsetAsync(d_goes_on + count, 0, sizeof(uint32_t), nullptr));
            BUILD_TRY(rocprim::exclusive_scan(d_scan_temp, scan_bytes, d_goes_on, d_place, 0u, (size_t)count + 1, rocprim::plus<uint32_t>(), nullptr));
            hipLaunchKernelGGL(compact_kernel, dim3(blocks_for(count)), dim3(BUILD_BLOCK), 0, nullptr, count, cl[1], cl[0], d_goes_on, d_place);
            BUILD_TRY(hipGetLastError());
            uint32_t left = 0, parts_left = 0;
            BUILD_TRY(hipMemcpy(&left, d_place + count, sizeof left, hipMemcpyDeviceToHost));
            BUILD_TRY(hipMemcpy(&parts_left, d_parts_left, sizeof parts_left, hipMemcpyDeviceToHost));
            if (!objects_join && !parts_left) {          // this round saw whole objects only (and joined nothing): from now on they join
                objects_join = true;
                continue;
            }
            // the pair with the smallest box of a round chose each other, so every round joins at least one pair
            if (left == 0 || left >= count) { err = "agglomeration stalled at " + std::to_string(count) + " clusters"; release_device_scene(out); return -1; }
            if (round >= PLOC_MAX_ROUNDS) { release_device_scene(out); return -2; }
            count = left;
        }
        {
            // what is left is the root, an inner node
            float4 top;
            BUILD_TRY(hipMemcpy(&top, cl[0].hi, sizeof top, hipMemcpyDeviceToHost));
            uint32_t top_ref;
            memcpy(&top_ref, &top.w, 4);
            BUILD_TRY(hipMemcpy(&inner, d_made, sizeof inner, hipMemcpyDeviceToHost));
            if ((top_ref & (LEAF_BIT | TRI_BIT)) || top_ref >= inner) { err = "agglomeration left no root"; release_device_scene(out); return -1; }
            root_node = top_ref;
            BUILD_TRY(rocprim::exclusive_scan(d_scan_temp, scan_bytes, b.leaf_size, b.leaf_pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), nullptr));
            uint32_t* d_packed = nullptr;
            BUILD_TRY(tmp.get(&d_packed, n));
            hipLaunchKernelGGL(packed_order_kernel, dim3(blocks_for(n)), dim3(BUILD_BLOCK), 0, nullptr, n, b, (const uint32_t*)d_ids_sorted, d_packed);
            BUILD_TRY(hipGetLastError());
            d_packed_ids = d_packed;
        }
    }
    hipLaunchKernelGGL(emit_triangles_kernel, dim3(blocks_for(n)), dim3(BUILD_BLOCK), 0, nullptr, d_tris, d_packed_ids, n, d_first, n_objects,
                       (TriIsect*)out.isect, (TriShade*)out.shade, (TriOrigin*)out.origin);
    BUILD_TRY(hipGetLastError());

    uint32_t n4 = 0, stack_need = 0, max_depth = 0;
    float root_box[6];
    uint32_t root_ref;
    if (inner == 0) {
        // one leaf: the scene root refers to it directly, there are no records
        BUILD_TRY(hipMalloc(&out.nodes4, sizeof(Node4)));
        BUILD_TRY(hipMemsetAsync(out.nodes4, 0, sizeof(Node4), nullptr));
        BUILD_TRY(hipMemcpy(root_box, b.leaf_box, sizeof root_box, hipMemcpyDeviceToHost));
        root_ref = REF_LEAF | 0u | (n > 1 ? REF_PAIR : 0u);
    } else {
        // ---- 5: even-depth inner nodes -> records, numbered by (level, falling area) ----
        uint32_t *d_even = nullptr, *d_even_pos = nullptr, *d_max_depth = nullptr, *d_level = nullptr, *d_front[2] = {nullptr, nullptr}, *d_front_count = nullptr;
        BUILD_TRY(tmp.get(&d_even, (size_t)inner + 1));
        BUILD_TRY(tmp.get(&d_even_pos, (size_t)inner + 1));
        BUILD_TRY(tmp.get(&d_max_depth, 1));
        BUILD_TRY(tmp.get(&d_level, inner));
        BUILD_TRY(tmp.get(&d_front[0], inner));
        BUILD_TRY(tmp.get(&d_front[1], inner));
        BUILD_TRY(tmp.get(&d_front_count, 1));
        BUILD_TRY(hipMemsetAsync(d_max_depth, 0, sizeof(uint32_t), nullptr));
        BUILD_TRY(hipMemsetAsync(d_even, 0, ((size_t)inner + 1) * sizeof(uint32_t), nullptr));
        hipLaunchKernelGGL(depth_kernel, dim3(blocks_for(inner)), dim3(BUILD_BLOCK), 0, nullptr, inner, b, d_max_depth);
        BUILD_TRY(hipGetLastError());
        // the records' roots, level after level from the top (a level's roots are the inner slots of the level above)
        {
            BUILD_TRY(hipMemcpy(d_front[0], &root_node, sizeof(uint32_t), hipMemcpyHostToDevice));
            uint32_t count = 1;
            for (uint32_t level = 0; count > 0; level++) {
                if (level > inner) { err = "the hierarchy has a cycle"; release_device_scene(out); return -1; }
                BUILD_TRY(hipMemsetAsync(d_front_count, 0, sizeof(uint32_t), nullptr));
                hipLaunchKernelGGL(mark_records_kernel, dim3(blocks_for(count)), dim3(BUILD_BLOCK), 0, nullptr, (const uint32_t*)d_front[level & 1], count, b, level, d_even,
                                   d_level, d_front[(level & 1) ^ 1], d_front_count);
                BUILD_TRY(hipGetLastError());
                BUILD_TRY(hipMemcpy(&count, d_front_count, sizeof count, hipMemcpyDeviceToHost));
            }
        }
        {
            size_t bytes = 0;
            BUILD_TRY(rocprim::exclusive_scan(nullptr, bytes, d_even, d_even_pos, 0u, (size_t)inner + 1, rocprim::plus<uint32_t>(), nullptr));
            char* d_temp = nullptr;
            BUILD_TRY(tmp.get(&d_temp, bytes));
            BUILD_TRY(rocprim::exclusive_scan(d_temp, bytes, d_even, d_even_pos, 0u, (size_t)inner + 1, rocprim::plus<uint32_t>(), nullptr));
        }
        BUILD_TRY(hipMemcpy(&n4, d_even_pos + inner, sizeof n4, hipMemcpyDeviceToHost));
        BUILD_TRY(hipMemcpy(&max_depth, d_max_depth, sizeof max_depth, hipMemcpyDeviceToHost));
        uint64_t *d_k4 = nullptr, *d_k4_sorted = nullptr;
        uint32_t *d_n4 = nullptr, *d_n4_sorted = nullptr, *d_rank = nullptr;
        BUILD_TRY(tmp.get(&d_k4, n4));
        BUILD_TRY(tmp.get(&d_k4_sorted, n4));
        BUILD_TRY(tmp.get(&d_n4, n4));
        BUILD_TRY(tmp.get(&d_n4_sorted, n4));
        BUILD_TRY(tmp.get(&d_rank, inner));
        hipLaunchKernelGGL(even_keys_kernel, dim3(blocks_for(inner)), dim3(BUILD_BLOCK), 0, nullptr, inner, b, d_even, d_even_pos, (const uint32_t*)d_level, d_k4, d_n4);
        BUILD_TRY(hipGetLastError());
        {
            size_t bytes = 0;
            BUILD_TRY(rocprim::radix_sort_pairs(nullptr, bytes, d_k4, d_k4_sorted, d_n4, d_n4_sorted, (size_t)n4, 0, 48, nullptr));
            char* d_temp = nullptr;
            BUILD_TRY(tmp.get(&d_temp, bytes));
            BUILD_TRY(rocprim::radix_sort_pairs(d_temp, bytes, d_k4, d_k4_sorted, d_n4, d_n4_sorted, (size_t)n4, 0, 48, nullptr));
        }
        hipLaunchKernelGGL(rank_kernel, dim3(blocks_for(n4)), dim3(BUILD_BLOCK), 0, nullptr, n4, d_n4_sorted, d_rank);
        // ---- 6: records, slot order, stack need ----
        Records rec = {};
        BUILD_TRY(hipMalloc(&out.nodes4, (size_t)n4 * sizeof(Node4)));
        rec.nodes4 = (Node4*)out.nodes4;
        BUILD_TRY(tmp.get(&rec.parent4, n4));
        BUILD_TRY(tmp.get(&rec.inner_slots, n4));
        BUILD_TRY(tmp.get(&rec.arrived4, n4));
        BUILD_TRY(tmp.get(&rec.need4, n4));
        BUILD_TRY(hipMemsetAsync(rec.arrived4, 0, (size_t)n4 * sizeof(uint32_t), nullptr));
        hipLaunchKernelGGL(emit_records_kernel, dim3(blocks_for(n4)), dim3(BUILD_BLOCK), 0, nullptr, n4, n, d_n4_sorted, d_rank, b, rec);
        hipLaunchKernelGGL(need_kernel, dim3(blocks_for(n4)), dim3(BUILD_BLOCK), 0, nullptr, n4, rec, 1);
        BUILD_TRY(hipGetLastError());
        BUILD_TRY(hipMemcpy(&stack_need, rec.need4, sizeof stack_need, hipMemcpyDeviceToHost));      // record 0 is the root
        BUILD_TRY(hipMemcpy(root_box, b.box + 6 * (size_t)root_node, sizeof root_box, hipMemcpyDeviceToHost));
        root_ref = 0u;
    }
    uint32_t* d_last = nullptr;
    BUILD_TRY(tmp.get(&d_last, 1));
    hipLaunchKernelGGL(last_triangle_kernel, dim3(1), dim3(1), 0, nullptr, (const Node4*)out.nodes4, root_ref, d_last);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipEventRecord(ev[2], nullptr));
    BUILD_TRY(hipMemcpy(&out.last_tri, d_last, sizeof(uint32_t), hipMemcpyDeviceToHost));
    BUILD_TRY(hipDeviceSynchronize());
    float ms_copy = 0.0f, ms_build = 0.0f;
    (void)hipEventElapsedTime(&ms_copy, ev[0], ev[1]);
    (void)hipEventElapsedTime(&ms_build, ev[1], ev[2]);
    for (int a = 0; a < 3; a++) { out.root.bmin[a] = root_box[a]; out.root.bmax[a] = root_box[3 + a]; }
    out.root.ref = root_ref;
    out.root.aux = 0;
    out.n_triangles = n;
    out.n_nodes4 = n4;
    out.stack_need = (int)stack_need;
    out.binary_depth = (int)max_depth + 1;
    out.bytes = ((size_t)n + 1) * sizeof(TriIsect) + (size_t)n * (sizeof(TriShade) + sizeof(TriOrigin)) + std::max<size_t>(n4, 1) * sizeof(Node4);
    out.copy_ms = ms_copy;
    out.build_ms = ms_build;
    return 0;
}

int build_scene_on_device(const rt_object* objects, int n_objects, DeviceScene& out, std::string& err) {
    const int rc = build_impl(objects, n_objects, out, err, true);
    if (rc != -2) return rc;
    std::fprintf(stderr, "rtcore: device build: the agglomeration needed more than %d rounds; building the hierarchy of the code bits instead\n", PLOC_MAX_ROUNDS);
    return build_impl(objects, n_objects, out, err, false);
}

}  // namespace rtc
