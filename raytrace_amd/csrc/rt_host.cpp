// rt_host.cpp -- host halves of the path: BVH build, vertex transform, camera
// frame, and the re-pack of upstream-layout arrays into the device layout.
// Built with -ffp-contract=off: every expression is plain IEEE binary32 in the
// operand order of the upstream code it replaces.
#include "rt_host.h"

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <queue>
#include <thread>

namespace rtc {

namespace {

struct V3 { float x, y, z; };

inline V3 ld(const rt_float3& p) { return V3{p.x, p.y, p.z}; }
inline float comp(const V3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

struct Box { V3 lo, hi; };

inline void grow(Box& b, const V3& p) {
    b.lo = V3{fminf(b.lo.x, p.x), fminf(b.lo.y, p.y), fminf(b.lo.z, p.z)};
    b.hi = V3{fmaxf(b.hi.x, p.x), fmaxf(b.hi.y, p.y), fmaxf(b.hi.z, p.z)};
}
inline void grow(Box& b, const Box& o) {
    b.lo = V3{fminf(b.lo.x, o.lo.x), fminf(b.lo.y, o.lo.y), fminf(b.lo.z, o.lo.z)};
    b.hi = V3{fmaxf(b.hi.x, o.hi.x), fmaxf(b.hi.y, o.hi.y), fmaxf(b.hi.z, o.hi.z)};
}
inline Box box_of(const rt_triangle& t) {
    Box b{ld(t.v[0].position), ld(t.v[0].position)};
    grow(b, ld(t.v[1].position));
    grow(b, ld(t.v[2].position));
    return b;
}
inline Box node_box(const rt_bvh_node& n) { return Box{ld(n.box_min), ld(n.box_max)}; }
inline void store_box(rt_bvh_node& n, const Box& b) {
    n.box_min = rt_float3{b.lo.x, b.lo.y, b.lo.z};
    n.box_max = rt_float3{b.hi.x, b.hi.y, b.hi.z};
}

// Worker threads for the host halves: RT_HOST_THREADS, else the machine's count capped at 16.
int host_threads() {
    if (const char* e = std::getenv("RT_HOST_THREADS")) {
        const int v = std::atoi(e);
        if (v >= 1) return v > 256 ? 256 : v;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw == 0 ? 1 : (hw > 16 ? 16 : int(hw));
}

// Nodes of the subtree over n triangles.  The split point depends on the count alone (the lower part takes
// (n-1)/2 + 1 triangles), so the shape of the tree, and with it the index of every node in the post-order array,
// is known before any triangle is looked at.
int subtree_nodes(int n) {
    if (n <= 2) return 1;
    const int lower = (n - 1) / 2 + 1, upper = n - lower;
    const int a = subtree_nodes(lower);
    return a + (upper == lower ? a : subtree_nodes(upper)) + 1;
}

// Median-split builder.  The permutation is carried by an index array; the
// comparator reads the centroid table, so std::nth_element sees exactly the
// comparison outcomes upstream's struct-array version sees (csrc/bvh.cpp:61-64)
// and leaves the same arrangement behind -- which decides the leaf pairing.
// Every subtree owns a range of the index array and a range of the output array, so subtrees are built by
// different threads without changing a byte of the result.
struct Builder {
    const rt_triangle* tris;
    std::vector<V3> centroid;
    std::vector<int> order;
    rt_bvh_node* out;

    struct Job { int first, last, base; };
    struct Top { int self, left, right; };               // nodes above the jobs, in creation order (parents first)
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Job> jobs;
    std::vector<Top> tops;
    int pending = 0;                                     // jobs queued or running
    int grain = 1 << 30;                                 // ranges up to this many triangles are built by one thread

    void emit_leaf(int first, int last, int at) {
        rt_bvh_node n;
        std::memset(&n, 0, sizeof(n));
        n.is_leaf = 1;
        n.left = order[first];
        n.right = -1;
        Box b = box_of(tris[n.left]);
        if (last > first) {
            n.right = order[first + 1];
            grow(b, box_of(tris[n.right]));
        }
        store_box(n, b);
        out[at] = n;
    }

    // Partition [first, last] at the median of the widest centroid axis; returns the centroid box.
    Box partition(int first, int last, int mid) {
        Box cb{V3{INFINITY, INFINITY, INFINITY}, V3{-INFINITY, -INFINITY, -INFINITY}};
        for (int i = first; i <= last; i++) grow(cb, centroid[order[i]]);
        int axis = 0;
        float widest = cb.hi.x - cb.lo.x;
        float w = cb.hi.y - cb.lo.y;
        if (w > widest) { axis = 1; widest = w; }
        w = cb.hi.z - cb.lo.z;
        if (w > widest) axis = 2;
        const V3* c = centroid.data();
        std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + last + 1,
                         [c, axis](int a, int b) { return comp(c[a], axis) < comp(c[b], axis); });
        return cb;
    }

    void finish_inner(int self, int left, int right, Box cb) {
        grow(cb, node_box(out[left]));
        grow(cb, node_box(out[right]));
        rt_bvh_node n;
        std::memset(&n, 0, sizeof(n));
        n.is_leaf = 0;
        n.left = left;
        n.right = right;
        store_box(n, cb);
        out[self] = n;
    }

    // inclusive range, csrc/bvh.cpp:29-75; the subtree's nodes go to out[base...], its root last; returns the root
    int split(int first, int last, int base) {
        if (first + 1 >= last) { emit_leaf(first, last, base); return base; }
        const int mid = (first + last) / 2;
        const Box cb = partition(first, last, mid);
        const int left = split(first, mid, base);
        const int right = split(mid + 1, last, left + 1);
        finish_inner(right + 1, left, right, cb);
        return right + 1;
    }

    // One queued range: small ones are built whole, large ones are partitioned and handed on as two ranges.  The
    // box of a handed-on node holds its centroid box until finish_tops() adds the boxes of its children.
    void run(const Job& j) {
        const int n = j.last - j.first + 1;
        if (n <= grain) { split(j.first, j.last, j.base); return; }
        const int mid = (j.first + j.last) / 2;
        const Box cb = partition(j.first, j.last, mid);
        const int left = j.base + subtree_nodes(mid - j.first + 1) - 1;
        const int right = left + subtree_nodes(j.last - mid);
        rt_bvh_node nd;
        std::memset(&nd, 0, sizeof(nd));
        store_box(nd, cb);
        out[right + 1] = nd;
        std::lock_guard<std::mutex> lock(mu);
        tops.push_back(Top{right + 1, left, right});
        jobs.push_back(Job{j.first, mid, j.base});
        jobs.push_back(Job{mid + 1, j.last, left + 1});
        pending += 2;
        cv.notify_all();
    }

    void worker() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv.wait(lock, [this] { return !jobs.empty() || pending == 0; });
                if (jobs.empty()) return;
                j = jobs.back();
                jobs.pop_back();
            }
            run(j);
            std::lock_guard<std::mutex> lock(mu);
            if (--pending == 0) cv.notify_all();
        }
    }

    void finish_tops() {                                  // children were created after their parents: walk backwards
        for (size_t i = tops.size(); i-- > 0;) {
            const Top& t = tops[i];
            finish_inner(t.self, t.left, t.right, node_box(out[t.self]));
        }
    }
};

}  // namespace

int build_bvh(const rt_triangle* tris, int n, std::vector<rt_bvh_node>& out) {
    out.clear();
    out.resize((size_t)subtree_nodes(n));
    Builder b;
    b.tris = tris;
    b.out = out.data();
    b.centroid.resize(n);
    b.order.resize(n);
    const float third = 1.f / 3.0f;   // upstream's float3 `/ 3.0f` multiplies by the reciprocal (common.cuh:148-153)
    for (int i = 0; i < n; i++) {
        const V3 a = ld(tris[i].v[0].position), bb = ld(tris[i].v[1].position), c = ld(tris[i].v[2].position);
        b.centroid[i] = V3{((a.x + bb.x) + c.x) * third, ((a.y + bb.y) + c.y) * third, ((a.z + bb.z) + c.z) * third};
        b.order[i] = i;
    }
    const int threads = n < (1 << 16) ? 1 : host_threads();
    if (threads <= 1) {
        b.split(0, n - 1, 0);
    } else {
        b.grain = std::max(1 << 13, n / (threads * 8));
        b.jobs.push_back(Builder::Job{0, n - 1, 0});
        b.pending = 1;
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; t++) pool.emplace_back([&b] { b.worker(); });
        b.worker();
        for (std::thread& t : pool) t.join();
        b.finish_tops();
    }
    return (int)out.size();
}

void apply_transform(const rt_transform& xf, rt_triangle* tris, int n) {
    // cuda/transforms.cu:47: degrees -> radians in double, then narrowed
    const float rx = (float)(xf.rotation.x * M_PI / 180.0f);
    const float ry = (float)(xf.rotation.y * M_PI / 180.0f);
    const float rz = (float)(xf.rotation.z * M_PI / 180.0f);
    const float cx = cosf(rx), sx = sinf(rx), cy = cosf(ry), sy = sinf(ry), cz = cosf(rz), sz = sinf(rz);
    // cuda/transforms.cu:11-19
    const float m[9] = {
        cy * cz,                 -cy * sz,                 sy,
        sx * sy * cz + cx * sz,  -sx * sy * sz + cx * cz,  -sx * cy,
        -cx * sy * cz + sx * sz, cx * sy * sz + sx * cz,   cx * cy};
    auto rot = [&m](rt_float3& v) {
        const float x = v.x, y = v.y, z = v.z;
        v.x = m[0] * x + m[1] * y + m[2] * z;
        v.y = m[3] * x + m[4] * y + m[5] * z;
        v.z = m[6] * x + m[7] * y + m[8] * z;
    };
    auto range = [&](int first, int last) {
        for (int t = first; t < last; t++)
            for (int k = 0; k < 3; k++) {
                rt_vertex& v = tris[t].v[k];
                v.position.x *= xf.scale.x; v.position.y *= xf.scale.y; v.position.z *= xf.scale.z;
                rot(v.position);
                v.position.x += xf.translation.x; v.position.y += xf.translation.y; v.position.z += xf.translation.z;
                rot(v.normal);
            }
    };
    const int threads = n < (1 << 16) ? 1 : host_threads();
    if (threads <= 1) { range(0, n); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back(range, int(int64_t(n) * t / threads), int(int64_t(n) * (t + 1) / threads));
    for (std::thread& t : pool) t.join();
}

CameraFrame camera_frame(int width, int height, const float pos[3], const float dir_in[3], const float up_in[3], float fov) {
    // cuda/camera.cu:49-59
    auto norm = [](V3 a) {
        const float len = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
        return V3{a.x / len, a.y / len, a.z / len};
    };
    auto crs = [](V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    V3 dir = norm(V3{dir_in[0], dir_in[1], dir_in[2]});
    V3 up = norm(V3{up_in[0], up_in[1], up_in[2]});
    const V3 right = crs(dir, up);
    up = crs(right, dir);
    const float aspect = (float)width / (float)height;
    // upstream writes an unqualified tan(fov / 2.0f) in a .cu file (cuda/camera.cu:55); with a float argument both
    // overload sets it can resolve to (CUDA's and <cmath>'s) give the binary32 function, so tanf it is.  Its last bit is
    // the host math library's; the oracle calls the same function.
    const float hh = tanf(fov / 2.0f);
    const float hw = aspect * hh;
    CameraFrame f;
    f.pos[0] = pos[0]; f.pos[1] = pos[1]; f.pos[2] = pos[2];
    // dir - hw*right - hh*up, where a - b is a + (-b)
    f.lower_left[0] = (dir.x + (-(right.x * hw))) + (-(up.x * hh));
    f.lower_left[1] = (dir.y + (-(right.y * hw))) + (-(up.y * hh));
    f.lower_left[2] = (dir.z + (-(right.z * hw))) + (-(up.z * hh));
    const float w2 = 2.0f * hw, h2 = 2.0f * hh;
    f.dx[0] = right.x * w2; f.dx[1] = right.y * w2; f.dx[2] = right.z * w2;
    f.dy[0] = up.x * h2; f.dy[1] = up.y * h2; f.dy[2] = up.z * h2;
    f.width = width;
    f.height = height;
    return f;
}

// ---------------------------------------------------------------------------
// Re-pack: upstream per-object arrays -> one global Node64 / TriIsect / TriShade set.
// ---------------------------------------------------------------------------
namespace {

// Emits one object's records.  The expanded size of every subtree (inner nodes, triangles) is counted first, so each
// subtree owns a range of the node array and of the triangle arrays; subtrees near the root go to threads of their own.
struct Packer {
    PackedScene* ps;
    const rt_object* obj;
    int obj_index;
    const uint32_t* inner_below;     // per upstream node: inner nodes of its subtree, itself included
    const uint32_t* tris_below;      // per upstream node: triangles of its subtree
    int spawn_depth;                 // subtrees above this depth are split over two threads

    static void put_box(ChildRec& c, const rt_bvh_node& n) {
        c.bmin[0] = n.box_min.x; c.bmin[1] = n.box_min.y; c.bmin[2] = n.box_min.z;
        c.bmax[0] = n.box_max.x; c.bmax[1] = n.box_max.y; c.bmax[2] = n.box_max.z;
        c.aux = 0;
    }

    void put_triangle(int ti, uint32_t at) const {
        const rt_triangle& t = obj->triangles[ti];
        TriIsect r;
        std::memset(&r, 0, sizeof(r));
        const rt_float3 &a = t.v[0].position, &b = t.v[1].position, &c = t.v[2].position;
        r.v0[0] = a.x; r.v0[1] = a.y; r.v0[2] = a.z;
        r.e1[0] = b.x + (-a.x); r.e1[1] = b.y + (-a.y); r.e1[2] = b.z + (-a.z);
        r.e2[0] = c.x + (-a.x); r.e2[1] = c.y + (-a.y); r.e2[2] = c.z + (-a.z);
        TriShade s;
        std::memset(&s, 0, sizeof(s));
        for (int k = 0; k < 3; k++) {
            float* n = k == 0 ? s.n0 : (k == 1 ? s.n1 : s.n2);
            float* uv = k == 0 ? s.uv0 : (k == 1 ? s.uv1 : s.uv2);
            n[0] = t.v[k].normal.x; n[1] = t.v[k].normal.y; n[2] = t.v[k].normal.z;
            uv[0] = t.v[k].texcoord[0]; uv[1] = t.v[k].texcoord[1];
        }
        s.material_id = t.material_id;
        ps->isect[at] = r;
        ps->shade[at] = s;
        ps->origin[at] = TriOrigin{obj_index, ti};
    }

    // Returns the reference for upstream node `ni`; its inner nodes go to nodes[node_at...], its triangles to
    // [tri_at...].  `deepest` receives the largest depth of an inner node below.
    uint32_t emit(int ni, int depth, uint32_t node_at, uint32_t tri_at, int& deepest) const {
        const rt_bvh_node& n = obj->bvh[ni];
        if (n.is_leaf) {
            uint32_t ref = REF_LEAF | tri_at;
            put_triangle(n.left, tri_at);
            if (n.right >= 0) { put_triangle(n.right, tri_at + 1); ref |= REF_PAIR; }
            return ref;
        }
        if (depth + 1 > deepest) deepest = depth + 1;
        const uint32_t me = node_at;
        // children are emitted right-first so that memory order follows upstream's visiting order
        const uint32_t right_nodes = me + 1, left_nodes = right_nodes + inner_below[n.right];
        const uint32_t right_tris = tri_at, left_tris = tri_at + tris_below[n.right];
        uint32_t rr, lr;
        if (depth < spawn_depth && tris_below[ni] >= (1u << 15)) {
            int deep_r = 0, deep_l = 0;
            std::thread other([&] { rr = emit(n.right, depth + 1, right_nodes, right_tris, deep_r); });
            lr = emit(n.left, depth + 1, left_nodes, left_tris, deep_l);
            other.join();
            deepest = std::max(deepest, std::max(deep_r, deep_l));
        } else {
            rr = emit(n.right, depth + 1, right_nodes, right_tris, deepest);
            lr = emit(n.left, depth + 1, left_nodes, left_tris, deepest);
        }
        Node64& out = ps->nodes[me];
        put_box(out.c[0], obj->bvh[n.left]);
        out.c[0].ref = lr;
        put_box(out.c[1], obj->bvh[n.right]);
        out.c[1].ref = rr;
        return me;
    }
};

}  // namespace

// Where to cut the object sequence [first, last).  Binary nodes at even depth below the root become two-level
// nodes (each absorbs its two children), so the best cut of a run depends on the parity of its depth.
struct TopSplits {
    int n = 0;
    std::vector<int> cut[2];                  // cut[depth & 1][first * (n + 1) + last]; empty = the middle
    int at(int first, int last, int depth) const {
        return cut[0].empty() ? (first + last) / 2 : cut[depth & 1][(size_t)first * (n + 1) + last];
    }
};

static ChildRec build_top(PackedScene& ps, const TopSplits& splits, int first, int last, int depth, int& max_depth) {
    if (last - first == 1) return ps.top[first];
    if (depth + 1 > max_depth) max_depth = depth + 1;
    const int mid = splits.at(first, last, depth);
    const uint32_t me = (uint32_t)ps.nodes.size();
    ps.nodes.emplace_back();
    const ChildRec lower = build_top(ps, splits, first, mid, depth + 1, max_depth);
    const ChildRec upper = build_top(ps, splits, mid, last, depth + 1, max_depth);
    Node64& n = ps.nodes[me];
    n.c[1] = lower;     // right child: visited first
    n.c[0] = upper;
    ChildRec r;
    for (int a = 0; a < 3; a++) {
        r.bmin[a] = lower.bmin[a] < upper.bmin[a] ? lower.bmin[a] : upper.bmin[a];
        r.bmax[a] = lower.bmax[a] > upper.bmax[a] ? lower.bmax[a] : upper.bmax[a];
    }
    r.ref = me;
    r.aux = 0;
    return r;
}

// Cuts for the object-level tree.  Any binary tree over contiguous runs of the object sequence keeps upstream's
// visiting order (the walk takes the lower run first), so the cuts are free to follow the geometry.  A ray enters a
// node about as often as the surface area of its box says (for rays without a preferred place or direction), and
// what it pays for is a visit of a two-level node, i.e. of a binary node at even depth; so the cuts minimise the
// summed box area of the even-depth runs, by dynamic programming over all runs and both parities.  A scene file
// lists, say, thirty parts of a car, then a ground plane and three lights: cutting in the middle puts half the car
// into one box with the ground, which every ray that looks down enters.  Falls back to the middle for very many
// objects or when the tree would become deep.
static void choose_top_splits(const PackedScene& ps, int n, TopSplits& out) {
    out = TopSplits();
    if (n < 3 || n > 384) return;
    const size_t w = (size_t)n + 1;
    std::vector<double> area(w * w, 0.0), cost[2] = {std::vector<double>(w * w, 0.0), std::vector<double>(w * w, 0.0)};
    std::vector<int> cut[2] = {std::vector<int>(w * w, 0), std::vector<int>(w * w, 0)};
    std::vector<int> depth[2] = {std::vector<int>(w * w, 0), std::vector<int>(w * w, 0)};
    auto finite_area = [](const float* lo, const float* hi) {
        const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        const double a = dx * dy + dy * dz + dz * dx;
        return (a == a && a < 1e300) ? a : 1e300;
    };
    for (int i = 0; i < n; i++) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int j = i + 1; j <= n; j++) {
            const ChildRec& c = ps.top[j - 1];
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c.bmin[a]); hi[a] = std::max(hi[a], c.bmax[a]); }
            area[i * w + j] = finite_area(lo, hi);
        }
    }
    for (int len = 2; len <= n; len++)
        for (int i = 0; i + len <= n; i++) {
            const int j = i + len, middle = (i + j) / 2;
            for (int parity = 0; parity < 2; parity++) {          // parity of this run's depth; its parts are one deeper
                const std::vector<double>& below = cost[parity ^ 1];
                double best = 1e308;
                int best_k = middle;
                for (int k = i + 1; k < j; k++) {
                    const double c = below[i * w + k] + below[k * w + j];
                    if (c < best || (c == best && std::abs(k - middle) < std::abs(best_k - middle))) { best = c; best_k = k; }
                }
                cost[parity][i * w + j] = (parity == 0 ? area[i * w + j] : 0.0) + best;
                cut[parity][i * w + j] = best_k;
                depth[parity][i * w + j] = 1 + std::max(depth[parity ^ 1][i * w + best_k], depth[parity ^ 1][best_k * w + j]);
            }
        }
    if (depth[0][0 * w + n] > 24) return;
    out.n = n;
    out.cut[0].swap(cut[0]);
    out.cut[1].swap(cut[1]);
}

// Collapse the binary node `bref` and its two children into one Node4; returns its index.
// `need` = stack entries a walk below this node can have pending at once.
static uint32_t collapse(PackedScene& ps, uint32_t bref, int& need) {
    const Node64& b = ps.nodes[bref];
    ChildRec slots[4];
    int used = 0;
    const ChildRec order[2] = {b.c[1], b.c[0]};          // right child is visited first
    for (const ChildRec& x : order) {
        if (x.ref & REF_LEAF) {
            slots[used++] = x;
        } else {
            slots[used++] = ps.nodes[x.ref].c[1];
            slots[used++] = ps.nodes[x.ref].c[0];
        }
    }
    const uint32_t me = (uint32_t)ps.nodes4.size();
    ps.nodes4.emplace_back();
    need = 0;
    for (int i = 0; i < used; i++) {
        int below = 0;
        if (!(slots[i].ref & REF_LEAF)) slots[i].ref = collapse(ps, slots[i].ref, below);
        // while the walk is below slot i, the later slots of this node may all be waiting on the stack
        const int pending = (used - 1 - i) + below;
        if (pending > need) need = pending;
    }
    Node4& out = ps.nodes4[me];
    std::memset(&out, 0, sizeof(out));
    for (int i = 0; i < 4; i++) {
        if (i < used) {
            for (int a = 0; a < 3; a++) { out.box[i][a] = slots[i].bmin[a]; out.box[i][3 + a] = slots[i].bmax[a]; }
            out.ref[i] = slots[i].ref;
        } else {
            for (int a = 0; a < 3; a++) { out.box[i][a] = INFINITY; out.box[i][3 + a] = -INFINITY; }
            out.ref[i] = REF_NONE;
        }
    }
    return me;
}

// Renumber the two-level nodes from the root.  Every walk starts at the root, so the nodes nearest to it are by
// far the most visited, and the kernels keep a prefix of the array in LDS.  Which nodes that prefix should hold:
// the ones a ray is most likely to enter -- for rays without a preferred place or direction that likelihood
// goes with the surface area of the node's box -- so the first RT_AREA_ORDERED numbers go to the node with the
// largest box among those whose parent is already numbered (a child's box is never larger than its parent's, so
// this is the set of the largest boxes); the rest follow breadth first.  Visiting order and results do not depend
// on node numbers.
#define RT_AREA_ORDERED 4096
static double box_area(const float* b) {      // min.xyz, max.xyz
    const double dx = (double)b[3] - b[0], dy = (double)b[4] - b[1], dz = (double)b[5] - b[2];
    const double a = dx * dy + dy * dz + dz * dx;
    return a == a ? a : 0.0;
}
static void renumber_from_root(PackedScene& ps) {
    const size_t n = ps.nodes4.size();
    std::vector<uint32_t> order;            // order[new] = old
    order.reserve(n);
    std::vector<uint32_t> renum(n, 0xFFFFFFFFu);
    order.push_back(ps.root.ref);
    renum[ps.root.ref] = 0;
    // largest box first, while the numbers that matter for the LDS prefix are being given out
    std::priority_queue<std::pair<double, uint32_t>> open;      // (area, old index) of nodes whose parent is numbered
    auto offer = [&](uint32_t numbered_old) {
        const Node4& nd = ps.nodes4[numbered_old];
        for (int k = 0; k < 4; k++) {
            const uint32_t r = nd.ref[k];
            if (r == REF_NONE || (r & REF_LEAF)) continue;
            open.push(std::make_pair(box_area(nd.box[k]), r));
        }
    };
    offer(ps.root.ref);
    while (!open.empty() && order.size() < RT_AREA_ORDERED) {
        const uint32_t r = open.top().second;
        open.pop();
        renum[r] = (uint32_t)order.size();
        order.push_back(r);
        offer(r);
    }
    // The nodes still waiting, then everything below them: breadth first, or with RT_NODE_ORDER=dfs each waiting node
    // followed at once by its whole subtree, depth first in visiting order (a subtree then owns one contiguous range of
    // the array, ever smaller ranges as the walk descends).  A record is exactly one cache line, so the order cannot
    // change what the caches hold, only which memory pages and DRAM rows a walk touches -- and that turned out not to
    // matter: 3223 against 3217 Mrays/s on the 30 M-triangle scene (4.4 GB), 8220 against 8110 on formula1.  Breadth
    // first stays the default; the other order stays selectable because measurements refer to it.
    const char* order_env = std::getenv("RT_NODE_ORDER");
    const bool breadth_first = !(order_env && std::string(order_env) == "dfs");
    if (breadth_first) {
        const size_t head0 = order.size();
        while (!open.empty()) {
            const uint32_t r = open.top().second;
            open.pop();
            renum[r] = (uint32_t)order.size();
            order.push_back(r);
        }
        for (size_t head = head0; head < order.size(); head++) {
            const Node4& nd = ps.nodes4[order[head]];
            for (int k = 0; k < 4; k++) {
                const uint32_t r = nd.ref[k];
                if (r == REF_NONE || (r & REF_LEAF)) continue;
                renum[r] = (uint32_t)order.size();
                order.push_back(r);
            }
        }
    } else {
        std::vector<uint32_t> stack;
        while (!open.empty()) {
            stack.push_back(open.top().second);
            open.pop();
            while (!stack.empty()) {
                const uint32_t r = stack.back();
                stack.pop_back();
                renum[r] = (uint32_t)order.size();
                order.push_back(r);
                const Node4& nd = ps.nodes4[r];
                for (int k = 3; k >= 0; k--) {               // slot 0 is visited first: it comes off the stack first
                    const uint32_t c = nd.ref[k];
                    if (c != REF_NONE && !(c & REF_LEAF)) stack.push_back(c);
                }
            }
        }
    }
    std::vector<Node4> out(order.size());
    for (size_t i = 0; i < order.size(); i++) {
        out[i] = ps.nodes4[order[i]];
        for (int k = 0; k < 4; k++) {
            uint32_t& r = out[i].ref[k];
            if (r != REF_NONE && !(r & REF_LEAF)) r = renum[r];
        }
    }
    ps.nodes4.swap(out);
    ps.root.ref = 0;
}

// A median-split tree over 2^30 triangles has 30 levels; the walk's stack (LDS + private spill) takes about 60.
#define RT_MAX_TREE_LEVELS 512
int pack_scene(const rt_object* objects, int n_objects, PackedScene& ps, std::string& err) {
    ps = PackedScene();
    for (int i = 0; i < n_objects; i++) {
        if (!objects[i].triangles || !objects[i].bvh || objects[i].n_triangles <= 0 || objects[i].n_bvh_nodes <= 0) {
            err = "object " + std::to_string(i) + ": empty triangle or node array";
            return -1;
        }
    }
    // Per object: validate child indices (a bad index would otherwise read out of bounds), and count what each
    // subtree expands to.  Children precede their parents in the array, so one forward pass does both.
    std::vector<std::vector<uint32_t>> inner_below(n_objects), tris_below(n_objects);
    uint64_t total_tris = 0, total_inner = 0;
    for (int i = 0; i < n_objects; i++) {
        const rt_object& o = objects[i];
        std::vector<uint32_t>& inner = inner_below[i];
        std::vector<uint32_t>& tri = tris_below[i];
        inner.resize(o.n_bvh_nodes);
        tri.resize(o.n_bvh_nodes);
        std::vector<uint16_t> levels(o.n_bvh_nodes);     // the emission below recurses once per level: a chain-shaped tree must not reach it
        for (int k = 0; k < o.n_bvh_nodes; k++) {
            const rt_bvh_node& n = o.bvh[k];
            const int lim = n.is_leaf ? o.n_triangles : o.n_bvh_nodes;
            if (n.left < 0 || n.left >= lim || n.right >= lim || (!n.is_leaf && (n.right < 0 || n.left >= k || n.right >= k))) {
                err = "object " + std::to_string(i) + ": node " + std::to_string(k) + " has an invalid child index";
                return -1;
            }
            if (n.is_leaf) {
                inner[k] = 0;
                tri[k] = n.right >= 0 ? 2 : 1;
                levels[k] = 1;
            } else {
                levels[k] = uint16_t(1 + std::max(levels[n.left], levels[n.right]));
                if (levels[k] > RT_MAX_TREE_LEVELS) {
                    err = "object " + std::to_string(i) + ": BVH deeper than " + std::to_string(RT_MAX_TREE_LEVELS) + " levels";
                    return -1;
                }
                const uint64_t a = uint64_t(inner[n.left]) + inner[n.right] + 1, b = uint64_t(tri[n.left]) + tri[n.right];
                if (a > REF_MASK || b > REF_MASK) { err = "scene exceeds 2^30-1 triangles"; return -1; }
                inner[k] = uint32_t(a);
                tri[k] = uint32_t(b);
            }
        }
        total_tris += tri[o.n_bvh_nodes - 1];
        total_inner += inner[o.n_bvh_nodes - 1];
    }
    if (total_tris > (uint64_t)REF_MASK || total_inner > (uint64_t)REF_MASK) { err = "scene exceeds 2^30-1 triangles"; return -1; }
    ps.isect.resize(total_tris);
    ps.shade.resize(total_tris);
    ps.origin.resize(total_tris);
    ps.nodes.reserve(total_inner + (size_t)n_objects);
    ps.nodes.resize(total_inner);
    ps.top.resize(n_objects);
    int spawn_depth = 0;
    for (int t = host_threads(); t > 1; t >>= 1) spawn_depth++;
    uint32_t node_at = 0, tri_at = 0;
    for (int i = 0; i < n_objects; i++) {
        const rt_object& o = objects[i];
        const Packer p{&ps, &o, i, inner_below[i].data(), tris_below[i].data(), spawn_depth};
        const int root = o.n_bvh_nodes - 1;   // cuda/raytrace.cu:49
        int deepest = 0;
        const uint32_t ref = p.emit(root, 0, node_at, tri_at, deepest);
        Packer::put_box(ps.top[i], o.bvh[root]);
        ps.top[i].ref = ref;
        if (deepest > ps.max_depth) ps.max_depth = deepest;
        node_at += inner_below[i][root];
        tri_at += tris_below[i][root];
    }
    // Object-level tree.  Upstream loops over the objects in ascending order and starts each
    // with a slab test of its root box (cuda/raytrace.cu:217-219, 49-55).  The same visiting order
    // falls out of the ordinary right-child-first walk when the objects are hung under a binary tree
    // over contiguous runs of the INDEX sequence whose right child holds the lower run: whole runs of
    // objects are skipped by one test, and every object still gets the test of its own root box.
    // A run's box is the union of its members' boxes, so it is hit whenever a member is and
    // its t_near never exceeds a member's (the slab arithmetic is monotone in the bounds).
    // Where the runs are cut is free; choose_top_splits() cuts where the boxes get small.
    int top_depth = 0;
    TopSplits splits;
    choose_top_splits(ps, n_objects, splits);
    ps.root = build_top(ps, splits, 0, n_objects, 0, top_depth);
    ps.max_depth = 0;
    if (!(ps.root.ref & REF_LEAF)) {
        int need = 0;
        ps.nodes4.reserve(ps.nodes.size() / 2 + 16);
        ps.root.ref = collapse(ps, ps.root.ref, need);
        ps.max_depth = need;
        const size_t before = ps.nodes4.size();
        renumber_from_root(ps);
        // every node kept its place in the tree under a new number: a reference out of range would send a kernel out of bounds
        bool sound = ps.nodes4.size() == before;
        for (const Node4& nd : ps.nodes4)
            for (int k = 0; k < 4; k++)
                if (nd.ref[k] != REF_NONE && !(nd.ref[k] & REF_LEAF) && nd.ref[k] >= ps.nodes4.size()) sound = false;
        if (!sound) { err = "internal: node renumbering lost a node"; return -1; }
    }
    // The triangle tested last by a walk that enters every box: always the slot visited last, then the leaf's last triangle.
    uint32_t ref = ps.root.ref;
    while (ref != REF_NONE && !(ref & REF_LEAF)) {
        const Node4& n = ps.nodes4[ref];
        uint32_t last = REF_NONE;
        for (int k = 0; k < 4; k++) if (n.ref[k] != REF_NONE) last = n.ref[k];
        ref = last;
    }
    ps.last_tri = ref == REF_NONE ? REF_NONE : (ref & REF_MASK) + ((ref & REF_PAIR) ? 1u : 0u);
    return 0;
}

}  // namespace rtc
