// rt_packed.h -- device-side scene layout, shared by the host re-packer and the kernels.
//
// Upstream keeps one 36-byte BVHNode array and one 104-byte Triangle array per OBJ shape (include/bvh.h:62-65,
// include/primitives.cuh:28-31) and fetches a node, tests its own box, then fetches children.  Here all objects hang
// under one tree, and what the kernels walk are 128-byte, 128-byte-aligned two-level records (Node4): the boxes of the
// up to four grandchildren of a binary node in visiting order, then their references -- one cache line decides two
// tree levels, and leaves need no node record at all (a leaf is a reference).  The 64-byte binary records (Node64)
// exist only while the packer builds the tree.  Triangles are split into a 48-byte intersection record (v0, e1, e2)
// stored in leaf order (a leaf's two triangles are adjacent, and the order of the array is upstream's visiting order)
// and a 64-byte shading record read once per hit.
#ifndef RT_PACKED_H
#define RT_PACKED_H

#include <stdint.h>

namespace rtc {

// Reference to a child: inner -> index into the node array (Node4 for the kernels); leaf -> packed
// triangle index of the leaf's first triangle, with flags.
constexpr uint32_t REF_LEAF = 0x80000000u;   // child is a leaf
constexpr uint32_t REF_PAIR = 0x40000000u;   // leaf holds two triangles (p and p+1)
constexpr uint32_t REF_MASK = 0x3FFFFFFFu;
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;   // "nothing to visit" marker inside the kernels

// One child slot: its box and its reference (32 bytes).
struct ChildRec {
    float bmin[3];
    float bmax[3];
    uint32_t ref;
    uint32_t aux;     // unused by the reference-order traversal
};

// Inner node: left child, right child (64 bytes).
struct alignas(64) Node64 {
    ChildRec c[2];
};

// Two-level node (128 bytes, one cache line): the up-to-four grandchildren of a binary node, in
// upstream's visiting order (slot 0 first).  A ray that hits a grandchild's box also hits the
// child's box in between (a parent box contains its children's and the slab arithmetic is monotone in
// the bounds), and the child's t_near never exceeds the grandchild's, so testing the grandchildren
// directly visits exactly the leaves upstream visits, in the same order -- with half as many
// dependent fetches per ray.  Unused slots carry ref = REF_NONE.
// Layout: the four boxes first (six floats each: min.x min.y min.z max.x max.y max.z -- every pair of floats
// the packed box test works on is 8-byte aligned), then the four references: a visit reads the first 112 bytes,
// seven 16-byte pieces per lane, not eight.
struct alignas(128) Node4 {
    float box[4][6];
    uint32_t ref[4];
    uint32_t pad[4];
};

// Intersection record (48 bytes): v0, e1 = v1 - v0, e2 = v2 - v0 -- the two edge
// subtractions of cuda/raytrace.cu:25-26 done once at pack time (same IEEE result).
struct alignas(16) TriIsect {
    float v0[3];
    float e1[3];
    float e2[3];
    uint32_t pad[3];
};

// Shading record (64 bytes): the per-vertex normals and texture coordinates and
// the material id that triangle_interpolate reads (cuda/raytrace.cu:80-96).
struct alignas(64) TriShade {
    float n0[3], n1[3], n2[3];
    float uv0[2], uv1[2], uv2[2];
    int32_t material_id;
};

// Where a packed triangle came from: upstream object index and triangle index.
struct TriOrigin {
    int32_t object;
    int32_t triangle;
};

struct MatDev {          // 32 bytes
    int32_t type;
    float albedo[3];
    float optical_density;
    float metal_fuzz;
    int32_t texture;
    int32_t pad;
};

struct TexDev {          // texels are RGBA8 packed in one uint32 pool
    uint32_t offset;     // first texel in the pool
    int32_t width, height;
    int32_t pad;
};

static_assert(sizeof(ChildRec) == 32, "ChildRec");
static_assert(sizeof(Node64) == 64, "Node64");
static_assert(sizeof(Node4) == 128, "Node4");
static_assert(sizeof(TriIsect) == 48, "TriIsect");
static_assert(sizeof(TriShade) == 64, "TriShade");
static_assert(sizeof(MatDev) == 32, "MatDev");

}  // namespace rtc
#endif
