// rt_kernels.hip -- gfx950 kernels of the path-tracing core and the C ABI around them.
//
// Kernel design (wave64, 256 CUs, no MFMA: nothing here is a dense contraction):
//  * Two schedules of the same paths, bit-identical in their results.  Wavefront (default): path state lives in HBM
//    pools and every iteration runs one kernel per stage -- wf_shade_kernel (consume hits, scatter, compact survivors
//    with 64-bit ballots), wf_plan_kernel + wf_generate_kernel (refill with new camera paths in tile order) and
//    wf_extend_kernel (closest hit).  Megakernel: path_kernel, one lane tracing a chunk of samples of one pixel as a
//    flat state machine.  Either way every finished path writes its radiance to a slot per (sample, pixel), and
//    accumulate_kernel sums the slots in upstream's order (cuda/raytrace.cu:208-246, csrc/worker.cpp:71-91).
//  * wf_extend_kernel fills the chip once per launch (768-lane workgroups, two per CU); waves claim chunks of 256 pool
//    entries from one queue per XCD, lanes take the next ray of the chunk as soon as theirs ends, and each trip the
//    wave runs the visit kind (node or leaf) most of its lanes wait for.
//  * The walk visits nodes in upstream's order (objects ascending, right child first, leaf left triangle then right,
//    prune on t_near > best t), which makes hit records bit-identical; it reads one 128-byte two-level record per
//    visit (the boxes of up to four grandchildren, rt_packed.h), hoists 1/dir out of the slab test, keeps the first
//    stack levels in LDS (one 8-byte column per lane, conflict-free) and the most-entered node records in LDS too.
//  * Random numbers are Philox4x32-10 with addressed draws (pixel, sample, slot, stream): no state array, results
//    independent of launch batching, kernel design and GPU count.
//
// Built with -ffp-contract=off; expressions keep upstream's operand order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rt_build.h"
#include "rt_device.h"

__device__ __forceinline__ void add_trav_counters(unsigned long long* counters, const TravCounters& tc) {
    wave_add(counters + RT_CNT_BOX_TESTS, tc.n_box);
    wave_add(counters + RT_CNT_TRI_TESTS, tc.n_tri);
    atomicMax(counters + RT_CNT_MAX_STACK, (unsigned long long)tc.max_sp);
    wave_add(counters + RT_CNT_WAVE_NODE_TRIPS, tc.w_node);
    wave_add(counters + RT_CNT_WAVE_LEAF_TRIPS, tc.w_leaf);
    wave_add(counters + RT_CNT_NODE_VISITS, tc.n_node);
    wave_add(counters + RT_CNT_HOT_VISITS, tc.n_hot_visit);
    wave_add(counters + RT_CNT_NONFINITE_RAYS, tc.n_nonfinite);
    atomicMax(counters + RT_CNT_MAX_RAY_VISITS, (unsigned long long)max(tc.max_ray_visits, tc.ray_visits));
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
    ShadeConst sk;
    sk.ambient = mk3(pp.ambient[0], pp.ambient[1], pp.ambient[2]);
    sk.rr = pp.rr; sk.k0 = k0; sk.k1 = k1;
    const v3 cam_pos = mk3(pp.cam.pos[0], pp.cam.pos[1], pp.cam.pos[2]);

    uint32_t n_rays = 0, n_shaded = 0, n_tex = 0, n_paths = 0;
    TravCounters tc = {};
    tc.hist = pp.hist;
    if (pp.hist && pp.hist_bands) tc.hist += (size_t)(ty * pp.hist_bands / pp.tiles_y) * RT_HIST_NODES;      // the band of this tile row

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
            closest_hit<COUNT>(sc, ray, h, stack, RT_BLOCK, pp.lds_slots, tc);
            const uint32_t bounce = (uint32_t)(pp.max_depth - 1 - depth);              // scatters done so far on this path
            finished = shade_hit<COUNT>(sc, sk, ray, h, (uint32_t)pixel, (uint32_t)sample, bounce, color, n_shaded, n_tex);
        }
        if (finished) {
            n_paths++;
            // a path that ran out of depth adds nothing (raytrace.cu:243): its slot holds +0, which is neutral
            // in the ordered sum because every other term is a product of non-negative factors
            if (!contributes) color = mk3(0.0f, 0.0f, 0.0f);
            float* slot = slot_of(pp, (uint32_t)pixel, sample);
            slot[0] = color.x; slot[1] = color.y; slot[2] = color.z;
            sample++;
            if (sample >= end) alive = false;
            else begin_path();
        }
    }

    wave_add(pp.counters + RT_CNT_RAYS, n_rays);
    wave_add(pp.counters + RT_CNT_WALKS, n_rays);
    wave_add(pp.counters + RT_CNT_PATHS, n_paths);
    if (COUNT) {
        wave_add(pp.counters + RT_CNT_SHADED, n_shaded);
        wave_add(pp.counters + RT_CNT_TEX_FETCHES, n_tex);
        add_trav_counters(pp.counters, tc);
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
// wavefront variant: path state in HBM, one kernel per stage
// ---------------------------------------------------------------------------
// The same paths as path_kernel (same addressed draws, same traversal, same slots), scheduled differently:
// a pool of up to `capacity` paths lives in HBM as three float4 planes; every iteration runs
//   shade    : consume the hit of every pooled path (miss / light / roulette / scatter), write finished paths'
//              radiance to their slots and COMPACT the survivors into the other pool with a 64-bit ballot +
//              lane prefix count + one atomic per wave;
//   plan     : one thread sizes the refill (free room vs. paths not yet started);
//   generate : append that many new camera paths, in tile order, so fresh primary rays fill whole waves;
//   extend   : closest hit for every pooled path.
// Every stage runs with full waves and few registers; the price is ~200 B of state traffic per segment.
// Pool entry: three 16-byte records.  A new path has only the first: its origin is the camera, its sample index follows from
// its place in the refill, its throughput is 1 and its depth the full depth.
#define RT_FRESH_PATH 0x40000000u   // in the pixel word: a path that has not been shaded yet
#define RT_PIXEL_MASK 0x03FFFFFFu
// in the pixel word, bits 26..29: this entry's ray is, bit for bit, the ray of the entry that many places before it in the
// pool (another sample of the same pixel that left a mirror or a pane of glass the same way): it is not walked, and its
// shading reads that entry's hit
#define RT_SAME_RAY_SHIFT 26
#define RT_SAME_RAY_MASK 0x3C000000u
struct WfPools {
    float4* p0[2];      // dir.xyz, pixel word: pixel | same-ray distance | RT_FRESH_PATH, or -1 when there is nothing to trace
    float4* p1[2];      // origin.xyz, sample (int bits); not stored for new paths
    float4* p2[2];      // color.rgb, depth left (int bits); not stored for new paths
    float4* hit;        // t, u, v, packed triangle (written by extend for the pool it ran on)
};

#define RT_QUEUES 8u
#define RT_QUEUE_STRIDE 32u
// The pool is cut into `regions` equal parts (a power of two, at most RT_REGIONS_MAX, each a multiple of 256 entries).
// A path stays in its region for life: the shading workgroups that read region r of one pool write their survivors to
// the front of region r of the other, each wave taking its places with one returning atomic on that region's own
// counter -- no counter is shared by workgroups that run at the same time (workgroup b reads region b mod regions), so
// the compaction needs neither a barrier nor a chip-wide hot word -- and the refill tops every region up with a
// contiguous run of new path ids.  The closest-hit launch sees the regions' live entries as one sequence
// (live_prefix) cut into RT_QUEUES queues, one per XCD.
#define RT_REGIONS_MAX 256u
#define RT_CURSOR_HISTORY 64u
#define RT_SUM_LOT_TILES 1024u      // completed tiles the ordered sum waits for before it runs (unless the slot ring is full)
struct WfCtl {
    // ---- what the host reads back (RT_CTL_HEAD_BYTES) ----
    uint32_t count_total[2];        // paths in pool 0 / 1 after the last refill of that pool
    uint32_t gen_count, regions;    // refill: number of new paths; regions of this pass
    unsigned long long cursor;      // next path id to start
    unsigned long long gen_first;   // first path id of the current refill
    uint32_t gen_rem;               // path id of the refill's first path, modulo the paths per tile ...
    uint32_t gen_tile;              // ... and divided by them: the tile the refill starts in
    // closest-hit launch under RT_OPT_COUNTERS: earliest wave start, latest wave end, sum of wave ends (100 MHz ticks), waves that did work
    unsigned long long t_first, t_last, t_end_sum, n_worked;
    uint32_t iter, last_gen_iter;   // refills planned so far in this pass; the last one of them that started new entries
    // ---- device only ----
    uint32_t queue_next[RT_QUEUES * RT_QUEUE_STRIDE];   // closest-hit launch: per queue, first entry no wave has claimed yet (one cache line each)
    uint32_t gen_base[RT_REGIONS_MAX];                  // per region: first place the current refill writes (= survivors in front of it)
    uint32_t gen_prefix[RT_REGIONS_MAX + 1];            // new paths the refill gives to the regions before r
    uint32_t live_prefix[RT_REGIONS_MAX + 1];           // live entries in the regions before r, after the refill
    uint32_t count[2][RT_REGIONS_MAX * RT_QUEUE_STRIDE];   // entries in region r of pool k (one cache line each)
    // the slot ring (PathParams::ring_tiles): the cursor after each of the last RT_CURSOR_HISTORY refills, the tiles whose
    // slots have been summed, and the range the ordered sum launched behind this refill is to take
    unsigned long long cursor_after[RT_CURSOR_HISTORY];
    uint32_t acc_done, acc_from, acc_to;
    uint32_t band;                  // which renumbered copy of the first node records the closest-hit launch reads (the band of tile rows now in flight)
};
#define RT_CTL_HEAD_BYTES 80
static_assert(offsetof(WfCtl, queue_next) == RT_CTL_HEAD_BYTES, "the host reads back exactly the head of the control block");

// One workgroup of RT_REGIONS_MAX threads, thread r for region r: sizes the refill of pool `nxt` (free room per region
// against the paths not yet started), hands out the new path ids region after region, and lays out the closest-hit
// launch's view of the pool.
// Slot ring: entries are started tile after tile and an entry started by refill j has ended with the shading launch of
// iteration j + `lifetime` (every shading launch takes one segment off the depth), so the tiles that lie wholly below the
// cursor of `lifetime` refills ago are complete: their slots are summed by the launch that follows this one
// (acc_from..acc_to) and handed on.  New entries are only started for tiles within `ring_tiles` of the last summed one.
// ring_tiles == 0: no ring, nothing is summed before the pass has drained.
__global__ __launch_bounds__(RT_REGIONS_MAX) void wf_plan_kernel(WfCtl* ctl, int nxt, unsigned long long total, uint32_t regions, uint32_t region_cap,
                                                                 uint32_t per_tile, uint32_t entry_yield, uint32_t ring_tiles, uint32_t lifetime,
                                                                 uint32_t tiles_x, uint32_t tiles_y, uint32_t n_bands, unsigned long long* counters) {
    __shared__ uint32_t scan[RT_REGIONS_MAX];
    const uint32_t r = threadIdx.x;
    if (r == 0 && ctl->n_worked) {
        // fold the previous closest-hit launch's wave times: slot-ticks the launch spanned, and those left idle behind waves that ended early
        // (atomic: with two pipelines per pass, two planners may run at once)
        atomicAdd(&counters[RT_CNT_LAUNCH_SLOT_TICKS], (ctl->t_last - ctl->t_first) * ctl->n_worked);
        atomicAdd(&counters[RT_CNT_TAIL_IDLE_TICKS], ctl->t_last * ctl->n_worked - ctl->t_end_sum);
    }
    const uint32_t have = r < regions ? ctl->count[nxt][r * RT_QUEUE_STRIDE] : 0u;
    // a new entry takes `entry_yield` lanes of the shading launch and may leave as many survivors behind (one per sample
    // of its run), all in this region of the other pool: only as many entries are started as the region has room for that
    // (the new entries' lanes start at a multiple of the run length, so that a run never straddles two waves of the shading launch)
    const uint32_t have_up = (have + entry_yield - 1u) / entry_yield * entry_yield;
    const uint32_t room = (r < regions && have_up < region_cap) ? (region_cap - have_up) / entry_yield : 0u;
    // exclusive prefix of the rooms: regions are filled in order until the paths run out
    auto exclusive = [&](uint32_t v) {
        scan[r] = v;
        __syncthreads();
        for (uint32_t d = 1; d < RT_REGIONS_MAX; d <<= 1) {
            const uint32_t up = r >= d ? scan[r - d] : 0u;
            __syncthreads();
            scan[r] += up;
            __syncthreads();
        }
        const uint32_t incl = scan[r];
        __syncthreads();
        return incl - v;
    };
    const uint32_t it_now = ctl->iter;
    const unsigned long long cursor_now = ctl->cursor;
    const uint32_t acc_done = ctl->acc_done;
    const uint32_t before = exclusive(room);
    // the room of all regions together: what this refill would start if nothing held it back
    __shared__ uint32_t all_room;
    if (r == RT_REGIONS_MAX - 1u) all_room = before + room;
    __syncthreads();
    uint32_t acc_to = acc_done;
    unsigned long long limit = total;
    if (ring_tiles) {
        uint32_t complete = acc_done;
        if (it_now >= lifetime) complete = (uint32_t)(ctl->cursor_after[(it_now - lifetime) % RT_CURSOR_HISTORY] / per_tile);
        if (complete < acc_done) complete = acc_done;
        // Completed tiles are summed in large lots (a launch over a few dozen tiles cannot keep HBM busy), unless the
        // ring is so full that waiting would hold this refill back.
        const unsigned long long wanted = cursor_now + (unsigned long long)all_room < total ? cursor_now + all_room : total;
        const bool ring_full = ((unsigned long long)acc_done + ring_tiles) * per_tile < wanted;
        const bool all_complete = (unsigned long long)complete * per_tile >= total;
        if (complete - acc_done >= RT_SUM_LOT_TILES || ring_full || all_complete) acc_to = complete;
        limit = ((unsigned long long)acc_to + ring_tiles) * per_tile;
        if (limit > total) limit = total;
    }
    const unsigned long long left = limit > cursor_now ? limit - cursor_now : 0ull;
    const uint32_t given = (unsigned long long)before >= left ? 0u : (uint32_t)(left - before < (unsigned long long)room ? left - before : room);
    const uint32_t new_before = exclusive(given);
    const uint32_t live_before = exclusive(have + given);
    __syncthreads();
    if (r < regions) {
        ctl->gen_base[r] = have;
        ctl->gen_prefix[r] = new_before;
        ctl->live_prefix[r] = live_before;
        ctl->count[nxt][r * RT_QUEUE_STRIDE] = have + given;
        ctl->count[nxt ^ 1][r * RT_QUEUE_STRIDE] = 0;     // the pool just consumed becomes the next target
    }
    if (r == regions - 1) {
        const uint32_t n = new_before + given;
        ctl->gen_prefix[regions] = n;
        ctl->live_prefix[regions] = live_before + have + given;
        ctl->count_total[nxt] = live_before + have + given;
        ctl->count_total[nxt ^ 1] = 0;
        ctl->gen_count = n;
        ctl->regions = regions;
        const uint32_t it = ctl->iter;
        if (n) ctl->last_gen_iter = it;
        ctl->iter = it + 1u;
        ctl->gen_first = ctl->cursor;
        ctl->gen_rem = (uint32_t)(ctl->cursor % per_tile);
        ctl->gen_tile = (uint32_t)(ctl->cursor / per_tile);
        ctl->cursor += n;
        ctl->cursor_after[it % RT_CURSOR_HISTORY] = ctl->cursor;
        ctl->acc_from = acc_done; ctl->acc_to = acc_to; ctl->acc_done = acc_to;
        // Most of the pool are entries of this refill and the one before: the tile row this refill starts in names the
        // band of the picture the launch works on (bands: n_bands equal shares of the tile rows, as the calibration cut
        // them).  Half a pool or a pool further back: 0.2 and 1.2 points fewer visits served from LDS.
        if (n_bands) {
            const uint32_t row = (uint32_t)min((unsigned long long)(tiles_y - 1u), cursor_now / per_tile / tiles_x);
            ctl->band = min(n_bands - 1u, row * n_bands / tiles_y);
        }
        ctl->t_first = ~0ull; ctl->t_last = 0; ctl->t_end_sum = 0; ctl->n_worked = 0;
    }
    if (r < RT_QUEUES) ctl->queue_next[r * RT_QUEUE_STRIDE] = 0;
}

// The same ordered sum for the tile layout of the wavefront variant (PathParams::ring_tiles): a workgroup takes a tile, a
// thread a pixel of it.  The tiles to sum come from the control block -- [acc_from, acc_to) as the refill planner set them,
// or with `rest` whatever is left up to `n_tiles` once the pass has drained.  Every pixel is summed once per pass, over
// all samples of the pass in order, so the additions and their order are those of accumulate_kernel.
__global__ __launch_bounds__(RT_BLOCK) void accumulate_tiles_kernel(const WfCtl* __restrict__ ctl, int rest, uint32_t n_tiles, uint32_t ring_tiles, int tiles_x,
                                                                    int width, int height, int first_sample, int n_samples, int batch, int flush,
                                                                    const float* __restrict__ slots, float* __restrict__ carry, float* __restrict__ out) {
    const uint32_t from = rest ? ctl->acc_done : ctl->acc_from, to = rest ? n_tiles : ctl->acc_to;
    for (uint32_t tile = from + blockIdx.x; tile < to; tile += gridDim.x) {
        const int tx = (int)(tile % (uint32_t)tiles_x), ty = (int)(tile / (uint32_t)tiles_x);
        const int x = tx * 16 + (int)(threadIdx.x & 15u), y = ty * 16 + (int)(threadIdx.x >> 4);
        if (x >= width || y >= height) continue;
        const size_t i = (size_t)y * (size_t)width + (size_t)x;
        const float* p = slots + ((size_t)(tile % ring_tiles) * (size_t)n_samples * 256u + threadIdx.x) * 3;
        float ox = out[3 * i], oy = out[3 * i + 1], oz = out[3 * i + 2];
        float tx_ = 0.0f, ty_ = 0.0f, tz_ = 0.0f;
        int in_batch = first_sample % batch;
        if (in_batch) { tx_ = carry[3 * i]; ty_ = carry[3 * i + 1]; tz_ = carry[3 * i + 2]; }
        // sixteen samples' loads are in flight before the first of them is added: a launch has few tiles (the ones
        // completed since the last refill), so the memory parallelism has to come from inside the thread
        for (int s = 0; s < n_samples; s += 16, p += 16 * 256 * 3) {
            float v[16][3];
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (s + k < n_samples) { v[k][0] = p[k * 768]; v[k][1] = p[k * 768 + 1]; v[k][2] = p[k * 768 + 2]; }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (s + k < n_samples) {
                    tx_ += v[k][0]; ty_ += v[k][1]; tz_ += v[k][2];
                    if (++in_batch == batch) {
                        ox += tx_; oy += ty_; oz += tz_;
                        tx_ = ty_ = tz_ = 0.0f;
                        in_batch = 0;
                    }
                }
            }
        }
        if (in_batch) {
            if (flush) { ox += tx_; oy += ty_; oz += tz_; }
            else { carry[3 * i] = tx_; carry[3 * i + 1] = ty_; carry[3 * i + 2] = tz_; }
        }
        out[3 * i] = ox; out[3 * i + 1] = oy; out[3 * i + 2] = oz;
    }
}

// The shading kernel streams the pools: it reads every record once and writes every survivor once, and a pool is far
// larger than the caches, so its accesses carry the non-temporal hint (+0.4 %; the same hint on the refill's writes and
// on the closest-hit kernel's ray reads and hit writes measured -0.2 %: those lines are still wanted soon after).
#ifndef RT_STREAM_NT
#define RT_STREAM_NT 1
#endif
__device__ __forceinline__ float4 pool_load(const float4* p) {
#if RT_STREAM_NT
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
__device__ __forceinline__ void pool_store(float4* p, const float4 v) {
#if RT_STREAM_NT
    f4v o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
    __builtin_nontemporal_store(o, reinterpret_cast<f4v*>(p));
#else
    *p = v;
#endif
}

// Path id -> (tile, sample of the pass, pixel of the 16x16 tile): ids walk a tile's 256 pixels, then the
// tile's next sample, then the next tile, so consecutive waves trace neighbouring primary rays.
// The samples of one semantic batch share their primary ray (cuda/raytrace.cu:210), so a thread does not take one path
// but one pixel's run of RT_GEN_RUN consecutive samples of a tile: the jitter draw, the two true divisions and the
// normalisation are done once per batch, and each of the run's paths costs a store.  (One thread per path spent 250
// vector instructions on every 16-byte record: 0.40 ms per refill where the writing takes 0.15.)
#define RT_GEN_RUN 64u
__global__ __launch_bounds__(RT_BLOCK) void wf_generate_kernel(const WfCtl* __restrict__ ctl, WfPools pools, int nxt, uint32_t regions, uint32_t log2_region_cap,
                                                               PathParams pp) {
    __shared__ uint32_t new_before[RT_REGIONS_MAX + 1];
    for (uint32_t k = threadIdx.x; k <= regions; k += RT_BLOCK) new_before[k] = ctl->gen_prefix[k];
    __syncthreads();
    uint32_t n_paths = 0;
    const uint32_t gen_count = ctl->gen_count, gen_rem = ctl->gen_rem;
    // the refill's ids, counted from the start of its first tile: [gen_rem, gen_rem + gen_count); the tiles they touch
    uint32_t unused;
    const uint32_t n_tiles = gen_count ? udiv_by(gen_rem + gen_count - 1u, pp.per_tile, pp.per_tile_magic, unused) + 1u : 0u;
    const uint32_t chunks_per_tile = ((uint32_t)pp.n_runs + RT_GEN_RUN - 1u) / RT_GEN_RUN;
    const uint32_t n_units = n_tiles * chunks_per_tile * 256u;         // fewer than the refill's entries plus two tiles' worth
    // the grid is sized for the chip, not for the pool: a workgroup strides over the units
    for (uint32_t u = blockIdx.x * RT_BLOCK + threadIdx.x; u < n_units; u += gridDim.x * RT_BLOCK) {
        const uint32_t p = u & 255u, chunk_index = u >> 8;
        const uint32_t tile_rel = chunk_index / chunks_per_tile, chunk = chunk_index - tile_rel * chunks_per_tile;
        const uint32_t tile = ctl->gen_tile + tile_rel;
        uint32_t tx_;
        const int ty = (int)udiv_by(tile, (uint32_t)pp.tiles_x, pp.tiles_x_magic, tx_), tx = (int)tx_;
        const int wave = p >> 6, lane = p & 63;
        const int x = tx * 16 + (wave & 1) * 8 + (lane & 7);
        const int y = ty * 16 + (wave >> 1) * 8 + (lane >> 3);
        const bool inside = x < pp.cam.width && y < pp.cam.height;
        const int pixel = inside ? y * pp.cam.width + x : 0;
        uint32_t reg = RT_REGIONS_MAX;       // the region that took this thread's last entry; none yet
        uint32_t batch_of_dir = 0xFFFFFFFFu;
        v3 dir = mk3(0.0f, 0.0f, 1.0f);
        const uint32_t k_end = min((uint32_t)pp.n_runs, (chunk + 1u) * RT_GEN_RUN);
        if (pp.group_log2) {
            // One entry per sample, a group's samples side by side (PathParams::group_log2): the 64 pixels of this wave
            // times the group's runs are one contiguous stretch of places.  The group lies inside one semantic batch, so
            // its entries of one pixel are one and the same record: every lane works out its own pixel's record once per
            // group, and the wave writes the stretch 64 consecutive places at a time, each lane fetching the record of the
            // pixel its place belongs to from the lane that made it.
            const uint32_t G = 1u << pp.group_log2, p0 = p & ~63u;
            for (uint32_t k0 = chunk * RT_GEN_RUN; k0 < k_end; k0 += G) {
                const uint32_t glen = min(G, k_end - k0), glen_magic = glen == G ? 0u : pp.group_rem_magic;
                int depth = pp.max_depth;
                if (inside) {
                    int s_lo, s_hi;
                    run_samples(pp, k0, s_lo, s_hi);
                    uint32_t in_batch;
                    const uint32_t batch_index = udiv_by((uint32_t)s_lo, (uint32_t)pp.batch, pp.batch_magic, in_batch);
                    if (batch_index != batch_of_dir) {
                        uint32_t w0, w1, w2, w3;
                        philox((uint32_t)pixel, batch_index, 0u, 1u, pp.key0, pp.key1, w0, w1, w2, w3);
                        dir = primary_dir(pp.cam, x, y, u01(w0), u01(w1));
                        batch_of_dir = batch_index;
                    }
                    if (depth == 0) {
                        for (int smp = s_lo; smp < s_lo + (int)glen; smp++) {
                            float* slot = slot_of(pp, (uint32_t)pixel, smp);
                            slot[0] = 0.0f; slot[1] = 0.0f; slot[2] = 0.0f;
                        }
                        depth = -1;
                        n_paths += glen;
                    }
                } else {
                    depth = -1;
                }
                const uint32_t word = depth > 0 ? ((uint32_t)pixel | RT_FRESH_PATH) : 0xFFFFFFFFu;
                const uint32_t stretch = tile_rel * pp.per_tile + place_of(pp, k0, p0);
                for (uint32_t i = 0; i < glen; i++) {                  // the same trip count in every lane: the exchanges below need the whole wave
                    const uint32_t e = i * 64u + (uint32_t)lane;
                    uint32_t within;
                    const uint32_t owner = glen == G ? e >> pp.group_log2 : udiv_by(e, glen, glen_magic, within);
                    const float ex = __shfl(dir.x, (int)owner, 64), ey = __shfl(dir.y, (int)owner, 64), ez = __shfl(dir.z, (int)owner, 64);
                    const uint32_t ew = (uint32_t)__shfl((int)word, (int)owner, 64);
                    const uint32_t along = stretch + e;
                    if (along < gen_rem) continue;
                    const uint32_t j = along - gen_rem;
                    if (j >= gen_count) continue;
                    if (reg == RT_REGIONS_MAX) {
                        uint32_t lo = 0, hi = regions - 1;
                        while (lo < hi) {
                            const uint32_t mid = (lo + hi + 1) >> 1;
                            if (new_before[mid] <= j) lo = mid; else hi = mid - 1;
                        }
                        reg = lo;
                    } else {
                        while (reg + 1 < regions && new_before[reg + 1] <= j) reg++;
                    }
                    const uint32_t pos = (reg << log2_region_cap) + ctl->gen_base[reg] + (j - new_before[reg]);
                    pools.p0[nxt][pos] = make_float4(ex, ey, ez, __uint_as_float(ew));
                }
            }
            continue;
        }
        for (uint32_t k = chunk * RT_GEN_RUN; k < k_end; k++) {
            const uint32_t along = tile_rel * pp.per_tile + place_of(pp, k, p);
            if (along < gen_rem) continue;                   // before the refill's first id (its first tile only)
            const uint32_t j = along - gen_rem;
            if (j >= gen_count) break;                       // past its last id (its last tile only)
            // the region that takes new entry j: the last one whose share starts at or before j -- searched for the
            // thread's first entry, stepped to for its later ones (j grows by 256 at a time)
            if (reg == RT_REGIONS_MAX) {
                uint32_t lo = 0, hi = regions - 1;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1) >> 1;
                    if (new_before[mid] <= j) lo = mid; else hi = mid - 1;
                }
                reg = lo;
            } else {
                while (reg + 1 < regions && new_before[reg + 1] <= j) reg++;
            }
            const uint32_t pos = (reg << log2_region_cap) + ctl->gen_base[reg] + (j - new_before[reg]);
            int depth = pp.max_depth;
            if (inside) {
                int s_lo, s_hi;
                run_samples(pp, k, s_lo, s_hi);              // never across a batch boundary: one jitter serves the run
                uint32_t in_batch;
                const uint32_t batch_index = udiv_by((uint32_t)s_lo, (uint32_t)pp.batch, pp.batch_magic, in_batch);
                if (batch_index != batch_of_dir) {
                    uint32_t w0, w1, w2, w3;
                    philox((uint32_t)pixel, batch_index, 0u, 1u, pp.key0, pp.key1, w0, w1, w2, w3);
                    dir = primary_dir(pp.cam, x, y, u01(w0), u01(w1));
                    batch_of_dir = batch_index;
                }
                if (depth == 0) {
                    // `while (depth--)` never runs: the paths add nothing (raytrace.cu:213, 243)
                    for (int smp = s_lo; smp < s_hi; smp++) {
                        float* slot = slot_of(pp, (uint32_t)pixel, smp);
                        slot[0] = 0.0f; slot[1] = 0.0f; slot[2] = 0.0f;
                    }
                    depth = -1;
                    n_paths += (uint32_t)(s_hi - s_lo);
                }
            } else {
                depth = -1;
            }
            pools.p0[nxt][pos] = make_float4(dir.x, dir.y, dir.z, __uint_as_float(depth > 0 ? ((uint32_t)pixel | RT_FRESH_PATH) : 0xFFFFFFFFu));
        }
    }
    wave_add(pp.counters + RT_CNT_PATHS, n_paths);
}

// Largest workgroup of wf_extend_kernel (its size is a launch parameter: the workgroups of a CU each keep their own
// LDS copy of the hot nodes, so fewer, larger workgroups leave room for more hot nodes).
#define RT_CNT_SPREAD 64u            // places the per-workgroup additions to RT_CNT_PATHS / RT_CNT_RAYS are spread over
#define RT_CNT_SPREAD_STRIDE 16u     // 64-bit words between two places: one 128-byte line each
#define RT_CNT_WORDS (RT_CNT_COUNT + RT_CNT_SPREAD * RT_CNT_SPREAD_STRIDE)
#define RT_EXTEND_BLOCK_MAX 1024
// Waves per SIMD the register allocator must leave room for in wf_extend_kernel (0 = its own choice).
#ifndef RT_EXTEND_WAVES
#define RT_EXTEND_WAVES 0
#endif
#if RT_EXTEND_WAVES > 0
#define RT_EXTEND_OCCUPANCY __attribute__((amdgpu_waves_per_eu(RT_EXTEND_WAVES, RT_EXTEND_WAVES)))
#else
#define RT_EXTEND_OCCUPANCY
#endif

// Closest hit for every pooled path.  Traversal lengths differ wildly between the rays of a wave, so a lane does
// not own one ray and a wave does not own a fixed share of the pool: the launch fills the chip once, every wave
// claims `chunk` consecutive pool entries at a time (one returning atomic per chunk, on the queue word of its XCD's
// region of the pool) and every lane that finishes a ray stores its hit and takes the next entry of its wave's
// chunk (64-bit ballot of the idle lanes, lane prefix count).  Lanes only run dry once, when the pool is drained.
// Which visit the wave runs next: VOTE false (`vote_leaf_weight` == 0) -> at most `node_burst` node trips, then one
// leaf trip; VOTE true (weight k > 0) -> whatever most lanes wait for, a leaf visit weighing k/16 node visits, and a
// refill once `refill_min` lanes are idle.  One ray's visits keep their order either way; only which lanes run
// together changes.
template <bool COUNT, bool VOTE>
__global__ __launch_bounds__(RT_EXTEND_BLOCK_MAX) RT_EXTEND_OCCUPANCY void wf_extend_kernel(SceneDev sc, WfCtl* ctl, WfPools pools, int nxt,
                                                             int lds_slots, uint32_t chunk, int node_burst, uint32_t n_hot,
                                                             int vote_leaf_weight, int refill_min, float cam_x, float cam_y, float cam_z,
                                                             uint32_t regions, uint32_t log2_region_cap, unsigned long long* counters,
                                                             const Node4* __restrict__ band_nodes, const uint32_t* __restrict__ band_roots, uint32_t top_n) {
    extern __shared__ __attribute__((aligned(16))) uint2 lds_stack[];
    lds_uint2* stack = (lds_uint2*)lds_stack + threadIdx.x;
    // top_n > 0: the first top_n node records are read from the copy numbered for the band of the picture in flight
    const uint32_t band = top_n ? ctl->band : 0u;
    const Node4* top = top_n ? band_nodes + (size_t)band * RT_HIST_NODES : sc.nodes;
    // behind the stacks, inside the one dynamic LDS block: copies of the first n_hot node records
    const uint32_t block = blockDim.x;
    lds_float4* hot = (lds_float4*)((lds_uint2*)lds_stack + (size_t)lds_slots * block);
    // behind those: where each region's live entries start in the launch's one sequence of entries (regions + 1 words)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* live = (lds_u32*)((__attribute__((address_space(3))) char*)hot + (size_t)n_hot * RT_HOT_STRIDE);
    for (uint32_t k = threadIdx.x; k <= regions; k += block) live[k] = ctl->live_prefix[k];
    const uint32_t count = ctl->count_total[nxt];
    TravCounters tc = {};
    for (uint32_t k = threadIdx.x; k < n_hot * 7u; k += block) {
        const uint32_t node = k / 7u, piece = k - node * 7u;
        const float4 v = reinterpret_cast<const float4*>(top)[node * 8u + piece];
        f4v o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
        hot[node * (RT_HOT_STRIDE / 16) + piece] = o;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint2 spill[RT_SPILL_SLOTS];
    Walker<COUNT> w;
    w.init(stack, (int)block, lds_slots, spill);
    w.set_hot(hot, n_hot);
    if (top_n) w.set_top(top, top_n, band_roots[band]);
    RayD r;
    ray_set(r, mk3(0.0f, 0.0f, 0.0f), mk3(0.0f, 0.0f, 1.0f));
    Hit h;
    h.t = RT_INF; h.u = 0.0f; h.v = 0.0f; h.tri = REF_NONE;
    int ridx = -1;                                 // pool entry this lane is tracing, -1 = none
    uint32_t next = 0, end = 0;                    // wave-uniform: the unclaimed rest of this wave's chunk (places in the sequence of live entries)
    uint32_t reg = 0;                              // wave-uniform: the pool region that holds entry `next`
    uint32_t queue = blockIdx.x & (RT_QUEUES - 1u);  // wave-uniform: the part of the sequence this wave draws from
    const uint32_t region = ((count + RT_QUEUES - 1u) / RT_QUEUES + 63u) & ~63u;
    unsigned long long cyc_claim = 0, n_claim = 0;
    bool exhausted = false;                        // wave-uniform: the queue has nothing left
    // hand the idle lanes `m` the next entries of the wave's chunk, claiming a new chunk when it is used up
    auto refill = [&](unsigned long long m, bool idle) {
        if (next >= end) {
            // the pool is cut into RT_QUEUES equal regions with one queue word each; a workgroup starts on the region of
            // its XCD (workgroups are dealt to the 8 XCDs in turn, so blockIdx & 7 names it) and moves on to the next
            // region for good when one is drained: each XCD's L2 sees one band of the picture until the very end
            bool got = false;
            for (uint32_t tries = 0; tries < RT_QUEUES && !got; tries++) {
                const uint32_t region_begin = min(count, queue * region), region_end = min(count, region_begin + region);
                const uint32_t size = chunk;
                unsigned long long tc0 = 0;
                if (COUNT) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tc0 = clock64(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ctl->queue_next[queue * RT_QUEUE_STRIDE], size);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (COUNT) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); cyc_claim += clock64() - tc0; n_claim++; }
                if (base < region_end - region_begin) {
                    next = region_begin + base;
                    end = min(region_end, next + size);
                    got = true;
                    // the pool region `next` lies in: the last one that starts at or before it (empty regions start where the next one does)
                    uint32_t lo = 0, hi = regions - 1u;
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi + 1u) >> 1;
                        if (live[mid] <= next) lo = mid; else hi = mid - 1u;
                    }
                    reg = lo;
                } else {
                    queue = (queue + 1u) & (RT_QUEUES - 1u);
                }
            }
            if (!got) { exhausted = true; return; }
        }
        // The lanes that take a new ray get it through selects at the end, not by assignments inside the branch: the ray and
        // the walk state then have one home register each for the whole trip loop (assigned in a branch, the new values
        // would live beside the old ones and every trip would end with some forty register copies).
        const uint32_t seq = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const bool take = idle && seq < end;
        // place in the sequence -> place in the pool: a chunk may run over the end of a region into the next non-empty one
        uint32_t rr = reg;
        while (take && seq >= live[rr + 1u]) rr++;
        const uint32_t idx = (rr << log2_region_cap) + (seq - live[rr]);
        float4 a = make_float4(0.0f, 0.0f, 1.0f, __uint_as_float(0xFFFFFFFFu));
        float4 b = make_float4(cam_x, cam_y, cam_z, 0.0f);          // a new path starts at the camera and has no second record
        if (take) {
            // both records are asked for at once, so that the second does not wait for the first one's flag (for a new
            // entry it is stale and dropped): one memory round trip per refill instead of two, -1.4 % kernel time
            a = pools.p0[nxt][idx];
            const float4 b1 = pools.p1[nxt][idx];
            if (!(__float_as_uint(a.w) & RT_FRESH_PATH)) b = b1;
        }
        // entries with nothing to trace (lanes outside the image) carry -1: their hit stays "none"
        // ... and so do entries whose ray another entry walks (their shading reads that entry's hit)
        const bool trace = take && (int)__float_as_uint(a.w) >= 0 && !(__float_as_uint(a.w) & RT_SAME_RAY_MASK);
        RayD rn;
        ray_set(rn, mk3(b.x, b.y, b.z), mk3(a.x, a.y, a.z));
        Hit hn;
        hn.t = RT_INF; hn.u = 0.0f; hn.v = 0.0f; hn.tri = REF_NONE;
        uint32_t first = REF_NONE;
        if (trace) first = w.first_ref(sc, rn, hn, tc);
        ridx = take ? (int)idx : ridx;
        h.t = take ? hn.t : h.t; h.u = take ? hn.u : h.u; h.v = take ? hn.v : h.v; h.tri = take ? hn.tri : h.tri;
        w.cur = trace ? first : w.cur;
        w.sp = trace ? 0 : w.sp;
        r.ox = trace ? rn.ox : r.ox; r.oy = trace ? rn.oy : r.oy; r.oz = trace ? rn.oz : r.oz;
        r.dx = trace ? rn.dx : r.dx; r.dy = trace ? rn.dy : r.dy; r.dz = trace ? rn.dz : r.dz;
        r.ix = trace ? rn.ix : r.ix; r.iy = trace ? rn.iy : r.iy; r.iz = trace ? rn.iz : r.iz;
        r.o_xy = trace ? rn.o_xy : r.o_xy; r.o_zx = trace ? rn.o_zx : r.o_zx; r.o_yz = trace ? rn.o_yz : r.o_yz;
        r.i_xy = trace ? rn.i_xy : r.i_xy; r.i_zx = trace ? rn.i_zx : r.i_zx; r.i_yz = trace ? rn.i_yz : r.i_yz;
        next = min(end, next + (uint32_t)__popcll(m));
        while (next < end && next >= live[reg + 1u]) reg++;
    };
    unsigned long long cyc_refill = 0, cyc_node = 0, cyc_leaf = 0;
    const unsigned long long cyc_begin = COUNT ? clock64() : 0ull;
    const unsigned long long tick_begin = COUNT ? wall_clock64() : 0ull;
    if (!VOTE) {
        for (;;) {
            const bool idle = w.done();
            if (idle && ridx >= 0) {
                pools.hit[ridx] = make_float4(h.t, h.u, h.v, __uint_as_float(h.tri));
                ridx = -1;
            }
            const unsigned long long m = __ballot(idle);
            if (m && !exhausted) refill(m, idle);
            if (!__any(ridx >= 0)) {
                if (exhausted) break;              // nothing in flight and nothing left to claim
                continue;
            }
            int burst = node_burst;
            while (w.at_inner()) {
                w.node_step(sc, r, h, tc);
                if (--burst == 0) break;
            }
            if (w.at_leaf()) w.leaf_step(sc, r, h, tc);
        }
    } else {
        // Two loops, so that the rays sit still in their registers while the wave walks: the outer one hands in finished
        // rays and takes new ones, the inner one runs trips until a refill is due (or nothing is left to visit).
        bool running = true;                       // wave-uniform
        while (running) {
            const bool idle = w.done();
            if (idle && ridx >= 0) {
                pools.hit[ridx] = make_float4(h.t, h.u, h.v, __uint_as_float(h.tri));
                ridx = -1;
            }
            const unsigned long long m = __ballot(idle);
            const bool any_walking = __any(!idle);
            if (m && !exhausted && ((int)__popcll(m) >= refill_min || !any_walking)) {
                const unsigned long long t0 = COUNT ? clock64() : 0ull;
                refill(m, idle);
                if (COUNT) cyc_refill += clock64() - t0;
            } else if (!any_walking) {
                running = false;                   // every lane has handed in its ray and there is nothing left to claim
            }
            for (;;) {
                const int n_idle = __popcll(__ballot(w.done()));
                const int n_node = __popcll(__ballot(w.at_inner())), n_leaf = __popcll(__ballot(w.at_leaf()));
                if (n_node + n_leaf == 0) break;                       // rays that ended at once are handed in above
                if (!exhausted && n_idle >= refill_min) break;         // a refill is due
                const unsigned long long t0 = COUNT ? clock64() : 0ull;
                if (n_node * 16 >= n_leaf * vote_leaf_weight) {
                    if (w.at_inner()) w.node_step(sc, r, h, tc);
                    if (COUNT) cyc_node += clock64() - t0;
                } else {
                    if (w.at_leaf()) w.leaf_step(sc, r, h, tc);
                    if (COUNT) cyc_leaf += clock64() - t0;
                }
            }
        }
    }
    if (COUNT) {
        add_trav_counters(counters, tc);
        if (lane == 0) {       // per wave: cycles inside each kind of trip, and from the first trip to the last
            atomicAdd(counters + RT_CNT_CYC_REFILL, cyc_refill);
            atomicAdd(counters + RT_CNT_CYC_NODE, cyc_node);
            atomicAdd(counters + RT_CNT_CYC_LEAF, cyc_leaf);
            atomicAdd(counters + RT_CNT_CYC_CLAIM, cyc_claim);
            atomicAdd(counters + RT_CNT_CLAIMS, n_claim);
            atomicAdd(counters + RT_CNT_CYC_WAVE, clock64() - cyc_begin);
            const unsigned long long tick_end = wall_clock64();
            atomicAdd(counters + RT_CNT_WAVE_SPAN_TICKS, tick_end - tick_begin);
            atomicAdd(counters + RT_CNT_WAVES, 1ull);
            if (tc.n_box) {          // this wave traced something
                atomicMin(&ctl->t_first, tick_begin);
                atomicMax(&ctl->t_last, tick_end);
                atomicAdd(&ctl->t_end_sum, tick_end);
                atomicAdd(&ctl->n_worked, 1ull);
            }
        }
    }
}

// Shading and queue compaction.  Workgroup b works on region (b mod regions) of pool `cur` -- workgroups that run at the
// same time read different regions -- one lane per (entry, sample): a continuing path is one lane, a NEW entry (a pixel's
// run of samples that share their primary ray and therefore its hit) as many lanes as a run of the pass can be long, next to each
// other, each shading the same hit with its own sample's draws.  The lanes of a run read the same pool and shading
// records (one cache line serves them), so the walk and its HBM traffic are spent once per run while the per-sample
// arithmetic stays what it is.  The planner admitted only as many new entries as leave `lanes <= region_cap`.
// Every wave files its survivors at the front of the same region of the other pool: 64-bit ballot, lane prefix count,
// ONE returning atomic per wave on that region's own counter.  No LDS, no barrier.  (A counter shared by the whole chip
// was the bottleneck of two earlier designs: a returning atomic on one word sustains ~90 per microsecond, and the
// barriers around a per-workgroup claim stalled 1024-lane workgroups as a whole.)
#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 256
#endif
template <bool COUNT>
__global__ __launch_bounds__(RT_SHADE_BLOCK) void wf_shade_kernel(SceneDev sc, WfCtl* ctl, WfPools pools, int cur, uint32_t regions, uint32_t log2_region_cap,
                                                                  PathParams pp, int share_equal) {
    const uint32_t reg = blockIdx.x % regions, within = blockIdx.x / regions;
    const uint32_t local = within * RT_SHADE_BLOCK + threadIdx.x;          // lane's place among the region's lanes
    const uint32_t live = ctl->count[cur][reg * RT_QUEUE_STRIDE];
    const uint32_t continuing = ctl->gen_base[reg];                          // the new entries lie behind the continuing ones
    // the new entries' lanes start at a multiple of the run length (a power of two up to 64 then keeps a run inside one wave)
    uint32_t over;
    (void)udiv_by(continuing, (uint32_t)pp.run_lanes, pp.run_lanes_magic, over);
    const uint32_t head = over ? continuing + (uint32_t)pp.run_lanes - over : continuing;
    const uint32_t lanes = head + (live - continuing) * (uint32_t)pp.run_lanes;
    if (within * RT_SHADE_BLOCK >= lanes) return;                            // nothing of this region is left for this workgroup
    const int nxt = cur ^ 1;
    uint32_t n_shaded = 0, n_tex = 0, n_paths = 0, n_rays = 0, n_walks = 0;
    bool survives = false, repeatable = false;
    float4 o0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), o1 = o0, o2 = o0;
    if (local < continuing || (local >= head && local < lanes)) {
        uint32_t entry = local, nth = 0;                                     // pool entry of the region; sample of its run
        if (local >= head) entry = continuing + udiv_by(local - head, (uint32_t)pp.run_lanes, pp.run_lanes_magic, nth);
        const uint32_t i = (reg << log2_region_cap) + entry;
        // a new entry: origin, throughput and depth are implied, and which run of samples it is follows from its place
        // in the refill, as wf_generate_kernel counted it out; any other path: second and third record.  Which of the two
        // an entry is follows from its place as well, so all of a path's records are asked for at once.
        float4 b = make_float4(pp.cam.pos[0], pp.cam.pos[1], pp.cam.pos[2], 0.0f);
        float4 c = make_float4(1.0f, 1.0f, 1.0f, __int_as_float(pp.max_depth));
        const float4 a = pool_load(&pools.p0[cur][i]);
        float4 hq = pool_load(&pools.hit[i]);
        if (entry < continuing) {
            b = pool_load(&pools.p1[cur][i]);
            c = pool_load(&pools.p2[cur][i]);
        }
        const uint32_t word = __float_as_uint(a.w);
        const uint32_t same_as = (int)word >= 0 ? (word & RT_SAME_RAY_MASK) >> RT_SAME_RAY_SHIFT : 0u;
        if (same_as) hq = pool_load(&pools.hit[i - same_as]);       // the hit of the entry that walked this ray
        if ((int)word >= 0) {                                        // -1: nothing was traced for this entry
            const int pixel = (int)(word & RT_PIXEL_MASK);
            int sample, s_hi;
            if (entry >= continuing) {
                uint32_t in_tile;
                (void)udiv_by(ctl->gen_rem + ctl->gen_prefix[reg] + (entry - continuing), pp.per_tile, pp.per_tile_magic, in_tile);
                int s_lo;
                run_samples(pp, run_of_place(pp, in_tile), s_lo, s_hi);
                sample = s_lo + (int)nth;
            } else {
                sample = __float_as_int(b.w);
                s_hi = sample + 1;
            }
            if (sample < s_hi) {                                     // a slice's last run may be shorter than the others
                int depth = __float_as_int(c.w);
                depth--;                                             // the segment extend just traced (raytrace.cu:213)
                n_rays = 1;
                n_walks = (nth == 0 && same_as == 0) ? 1u : 0u;
                ShadeConst sk;
                sk.ambient = mk3(pp.ambient[0], pp.ambient[1], pp.ambient[2]);
                sk.rr = pp.rr; sk.k0 = pp.key0; sk.k1 = pp.key1;
                RayD ray;
                ray.ox = b.x; ray.oy = b.y; ray.oz = b.z; ray.dx = a.x; ray.dy = a.y; ray.dz = a.z;
                ray.ix = ray.iy = ray.iz = 0.0f;                     // shading never reads the inverses
                Hit h;
                h.t = hq.x; h.u = hq.y; h.v = hq.z; h.tri = __float_as_uint(hq.w);
                v3 color = mk3(c.x, c.y, c.z);
                const uint32_t bounce = (uint32_t)(pp.max_depth - 1 - depth);
                bool finished = shade_hit<COUNT>(sc, sk, ray, h, (uint32_t)pixel, (uint32_t)sample, bounce, color, n_shaded, n_tex, repeatable);
                if (!finished && depth == 0) {
                    // the next `while (depth--)` test fails: out of depth, the path adds nothing (raytrace.cu:243)
                    finished = true;
                    color = mk3(0.0f, 0.0f, 0.0f);
                }
                if (finished) {
                    n_paths = 1;
                    float* slot = slot_of(pp, (uint32_t)pixel, sample);
                    slot[0] = color.x; slot[1] = color.y; slot[2] = color.z;
                } else {
                    survives = true;
                    o0 = make_float4(ray.dx, ray.dy, ray.dz, __int_as_float(pixel));
                    o1 = make_float4(ray.ox, ray.oy, ray.oz, __int_as_float(sample));
                    o2 = make_float4(color.x, color.y, color.z, __int_as_float(depth));
                }
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(survives);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));      // survivors in the lanes before this one
    // Samples that reached this hit along one ray and leave a mirror or a pane of glass the same way are, bit for bit, one
    // ray again: only the first of them (the lowest lane) needs to be walked.  The samples of a pixel's batch lie in
    // neighbouring lanes (they were filed together), so a lane looks at the 15 lanes before it, by a hash of pixel and
    // ray first and then word for word; the others carry the distance to the first one's place in the pool.
    const bool candidate = survives && repeatable && share_equal;
    if (__any(candidate)) {
        const uint32_t w[7] = {__float_as_uint(o0.w), __float_as_uint(o1.x), __float_as_uint(o1.y), __float_as_uint(o1.z),
                               __float_as_uint(o0.x), __float_as_uint(o0.y), __float_as_uint(o0.z)};
        uint32_t hash = w[0] * 0x9E3779B1u;
#pragma unroll
        for (int k = 1; k < 7; k++) hash = (hash ^ w[k]) * 0x85EBCA6Bu + (hash >> 15);
        int back = 0;
#pragma unroll
        for (int s = 1; s <= 15; s++) {
            const uint32_t their_hash = (uint32_t)__shfl_up((int)hash, s, 64);
            const int theirs = __shfl_up((int)candidate, s, 64);
            if (candidate && lane >= s && theirs && their_hash == hash) back = s;            // ends at the lowest lane that matches
        }
        int first = lane - back;
        bool same = back != 0;
#pragma unroll
        for (int k = 0; k < 7; k++) same = same & ((uint32_t)__shfl((int)w[k], first, 64) == w[k]);
        if (!same) first = lane;
        // runs longer than 16 samples: the lane found may itself have found an earlier one
        for (int hop = 0; hop < 4; hop++) {
            const int further = __shfl(first, first, 64);
            if (!__any(further != first)) break;
            first = further;
        }
        const uint32_t distance = rank - (uint32_t)__shfl((int)rank, first, 64);
        if (first != lane && distance >= 1u && distance <= (RT_SAME_RAY_MASK >> RT_SAME_RAY_SHIFT))
            o0.w = __uint_as_float(__float_as_uint(o0.w) | (distance << RT_SAME_RAY_SHIFT));
    }
    uint32_t base = 0;
    if (lane == 0 && m) base = atomicAdd(&ctl->count[nxt][reg * RT_QUEUE_STRIDE], (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (survives) {
        const uint32_t pos = (reg << log2_region_cap) + base + rank;
        pool_store(&pools.p0[nxt][pos], o0);
        pool_store(&pools.p1[nxt][pos], o1);
        pool_store(&pools.p2[nxt][pos], o2);
    }
    // finished paths, rays as upstream counts them (one per sample and depth-loop iteration, cuda/raytrace.cu:213-219) and
    // closest-hit walks actually run (one per entry): every wave adds its sums, spread over RT_CNT_SPREAD cache lines --
    // on one word the additions would queue up behind each other
    const uint32_t fin = (uint32_t)__popcll(__ballot(n_paths != 0)), rays = (uint32_t)__popcll(__ballot(n_rays != 0)),
                   walks = (uint32_t)__popcll(__ballot(n_walks != 0));
    if (lane == 0) {
        unsigned long long* spread = pp.counters + RT_CNT_COUNT + ((blockIdx.x * (RT_SHADE_BLOCK / 64) + (threadIdx.x >> 6)) & (RT_CNT_SPREAD - 1u)) * RT_CNT_SPREAD_STRIDE;
        if (fin) atomicAdd(spread, (unsigned long long)fin);
        if (rays) atomicAdd(spread + 1, (unsigned long long)rays);
        if (walks) atomicAdd(spread + 2, (unsigned long long)walks);
    }
    if (COUNT) {
        wave_add(pp.counters + RT_CNT_SHADED, n_shaded);
        wave_add(pp.counters + RT_CNT_TEX_FETCHES, n_tex);
    }
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
    uint32_t n_r = 0;
    TravCounters tc = {};
    if (i < n_rays) {
        RayD r;
        ray_set(r, mk3(rays[6 * (size_t)i], rays[6 * (size_t)i + 1], rays[6 * (size_t)i + 2]),
                mk3(rays[6 * (size_t)i + 3], rays[6 * (size_t)i + 4], rays[6 * (size_t)i + 5]));
        Hit h;
        n_r = 1;
        closest_hit<true>(sc, r, h, stack, RT_BLOCK, lds_slots, tc);
        int o = -1, ti = -1;
        if (h.tri != REF_NONE) { const TriOrigin og = sc.origin[h.tri]; o = og.object; ti = og.triangle; }
        obj[i] = o; tri[i] = ti; t[i] = h.t;
        if (u) u[i] = h.u;
        if (v) v[i] = h.v;
    }
    wave_add(counters + RT_CNT_RAYS, n_r);
    wave_add(counters + RT_CNT_WALKS, n_r);
    add_trav_counters(counters, tc);
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

// Renumbering of the first `k` node records (the candidates for the closest-hit kernel's LDS copy): record i moves to
// place perm[i]; references among the first k follow.  No record beyond k refers to one of the first k: parents always
// carry lower numbers than their children.
__global__ __launch_bounds__(RT_BLOCK) void permute_nodes_kernel(const Node4* __restrict__ nodes, uint32_t k, const uint32_t* __restrict__ perm, Node4* __restrict__ out) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= k) return;
    Node4 nd = nodes[i];
    for (int slot = 0; slot < 4; slot++) {
        const uint32_t r = nd.ref[slot];
        if (r != REF_NONE && !(r & REF_LEAF) && r < k) nd.ref[slot] = perm[r];
    }
    out[perm[i]] = nd;
}

// The same for several numberings at once: copy blockIdx.y of the first k records, numbered by perms[blockIdx.y][.],
// RT_HIST_NODES records apart in `out`; the array itself is left as it is.
__global__ __launch_bounds__(RT_BLOCK) void permute_bands_kernel(const Node4* __restrict__ nodes, uint32_t k, const uint32_t* __restrict__ perms, Node4* __restrict__ out) {
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= k) return;
    const uint32_t* perm = perms + (size_t)blockIdx.y * RT_HIST_NODES;
    Node4 nd = nodes[i];
    for (int slot = 0; slot < 4; slot++) {
        const uint32_t r = nd.ref[slot];
        if (r != REF_NONE && !(r & REF_LEAF) && r < k) nd.ref[slot] = perm[r];
    }
    out[(size_t)blockIdx.y * RT_HIST_NODES + perm[i]] = nd;
}

// ---------------------------------------------------------------------------
// context and C ABI
// ---------------------------------------------------------------------------
// Sample slots of rt_raytrace_static: one array per GPU the library runs on, held until the process ends.  In host
// code the array's name is only a handle for hipGetSymbolAddress; the device address each GPU gives its copy is looked
// up once per device in rt_create and kept in g_static_slots[device], which is what kernels and copies are given.
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];
#define RT_MAX_DEVICES 64
static float* g_static_slots[RT_MAX_DEVICES];

static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) { g_last_error = msg; return code; }

// No C++ exception may leave through the C ABI (a caller in C, Go or Python could not catch it): the entry points that
// allocate host memory in proportion to the scene, or start threads, run their bodies through this.
template <class Body>
static int guarded(const char* who, Body&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(RT_ERR_HOST, std::string(who) + ": out of host memory");
    } catch (const std::exception& e) {
        return fail(RT_ERR_HOST, std::string(who) + ": " + e.what());
    } catch (...) {
        return fail(RT_ERR_HOST, std::string(who) + ": unknown exception");
    }
}

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
    int opt_hot_tune = 2;            // wavefront: number the node records by measured visits before a frame (one calibration sample per pixel); 2 = per band of the picture as well
    bool hot_dirty = true;           // scene or camera changed since the node numbering was last tuned
    void* d_tune = nullptr;          // calibration scratch: visit counts, throw-away counters, permutation, a copy of the candidate records
    // RT_OPT_HOT_TUNE = 2: one renumbered copy of the first node records per band of the picture (the calibration counts
    // visits band by band), so that the records in LDS are the ones the rays in flight enter most
    Node4* d_band_nodes = nullptr;   // [RT_BANDS_MAX][RT_HIST_NODES]
    uint32_t* d_band_roots = nullptr;   // the root's number in each copy
    uint32_t n_bands = 0;            // copies that are valid for the present scene and camera (0: none)
    uint32_t band_records = 0;       // records in each copy
    int opt_share_equal = 1;         // wavefront: samples of a pixel that leave a mirror or glass along one and the same ray share that ray's walk too
    int opt_share_primary = 1;       // wavefront: one closest-hit query per pixel and semantic batch (its samples share the primary ray)
    int opt_gpu_bvh = 0;             // 1 = rt_upload_scene builds its own tree on the device instead of re-packing the caller's
    int opt_lds_slots = 11;          // stack levels kept in LDS (deeper ones spill to private memory)
    int opt_chunk = 8;               // samples one lane traces in a row
    int opt_pass = 1024;             // samples per pixel held in the slot buffer at once (less when that would exceed RT_SLOTS_MAX_BYTES)
    int opt_timing = 0;              // record HIP events around the traversal kernels
    std::vector<hipEvent_t> ev_pool; // event pairs: [2k] before, [2k+1] after
    std::vector<int> ev_kind;        // per pair: 0 = wf_extend, 1 = wf_shade, 2 = path_kernel
    size_t ev_used = 0;              // pairs recorded since the last collection
    double time_ms[3] = {0, 0, 0};
    uint64_t time_launches[3] = {0, 0, 0};
    int opt_mode = 1;                // 0 = megakernel (path_kernel), 1 = wavefront (stage kernels)
    int opt_wf_log2 = 27;            // wavefront pool capacity = 2^this paths
    int opt_wf_run = 256;            // pool entries a closest-hit wave claims from the queue at a time
    int opt_wf_sort = 1;             // shade kernel files surviving rays by direction octant
    int opt_wf_vote = 32;            // 0 = fixed node burst; k > 0 = per-trip vote, a leaf visit weighs k/16 node visits
    int opt_wf_refill = 24;          // vote policy: idle lanes that trigger a refill
    size_t extend_lds_limit = 64 * 1024;   // dynamic LDS the extend kernel may currently be launched with
    int opt_wf_block = 768;          // lanes per workgroup of the extend kernel: two workgroups of twelve waves per CU
    int opt_hot_nodes = 100;         // node records (breadth-first prefix) the extend kernel keeps in LDS
    int opt_wf_burst = 4;            // inner-node trips between two looks at leaves / idle lanes
    int n_cu = 256;                  // compute units of the device
    // The wavefront variant runs up to RT_WF_LANES_MAX independent pipelines ("lanes" of the host loop, nothing to do
    // with SIMD lanes), each with its own pools, control block and stream: while one pipeline's closest-hit launch
    // waits for its last long rays, the other pipeline's kernels fill the CUs that have already gone idle.
    struct WfLane {
        WfPools pools = {};
        WfCtl* d_ctl = nullptr;
        WfCtl* h_ctl = nullptr;      // pinned ring of control-block snapshots
        hipEvent_t ctl_ev[4] = {nullptr, nullptr, nullptr, nullptr};
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
    };
    WfLane lanes[2];
    hipEvent_t fork_ev = nullptr;
    int lanes_allocated = 0;
    int opt_wf_lanes = 1;            // pipelines per pass (2 measured slower: the second pipeline's shading starves behind the first's closest-hit launch)
    uint32_t pool_capacity = 0;
    uint32_t pool_cap_limit = 0xFFFFFFFFu;   // the largest pool this device had memory for, once an allocation has failed
    float* d_slots = nullptr;        // the sample slots of a pass (layouts: PathParams::ring_tiles), then one [n_pixel][3] carry plane
    size_t slots_bytes = 0;
    int opt_group_samples = 1;                    // wavefront, one walk per sample: the samples of a batch in neighbouring lanes
    size_t slots_ring_bytes = (size_t)10 << 30;   // wavefront: what the ring of tile slots may take (less for a small frame: never more than the frame's tiles)
    int stack_slots = 1;
    uint64_t stats[8] = {0};
    // rt_raytrace_start / rt_raytrace_wait: the call a library thread is driving, and how it ended
    std::thread job;
    int job_status = RT_OK;
    std::string job_error;
};

static void free_scene(rt_context* c) {
    void** ptrs[] = {&c->d_nodes, &c->d_isect, &c->d_shade, &c->d_origin, &c->d_mats, &c->d_tex, &c->d_texels};
    for (void** p : ptrs) { if (*p) { (void)hipFree(*p); *p = nullptr; } }
    c->have_scene = false;
}

static void free_pools(rt_context* c) {
    for (rt_context::WfLane& l : c->lanes) {
        for (int k = 0; k < 2; k++) {
            if (l.pools.p0[k]) (void)hipFree(l.pools.p0[k]);
            if (l.pools.p1[k]) (void)hipFree(l.pools.p1[k]);
            if (l.pools.p2[k]) (void)hipFree(l.pools.p2[k]);
            l.pools.p0[k] = l.pools.p1[k] = l.pools.p2[k] = nullptr;
        }
        if (l.pools.hit) (void)hipFree(l.pools.hit);
        l.pools.hit = nullptr;
        if (l.d_ctl) (void)hipFree(l.d_ctl);
        if (l.h_ctl) (void)hipHostFree(l.h_ctl);
        for (auto& e : l.ctl_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        l.d_ctl = nullptr; l.h_ctl = nullptr;
        if (l.done) (void)hipEventDestroy(l.done);
        l.done = nullptr;
        if (l.stream) (void)hipStreamDestroy(l.stream);
        l.stream = nullptr;
    }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    c->fork_ev = nullptr;
    c->lanes_allocated = 0;
    c->pool_capacity = 0;
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
int rt_set_last_error_(int code, const char* message) { return fail(code, message ? message : ""); }   // for the library's other source files

int rt_get_device(rt_context* ctx, int* device) {
    if (!ctx || !device) return fail(RT_ERR_INVALID, "rt_get_device: null argument");
    *device = ctx->device;
    return RT_OK;
}
int rt_abi_version(void) { return RTCORE_ABI_VERSION; }

int rt_build_bvh(const rt_triangle* triangles, int n_triangles, rt_bvh_node* nodes_out, int capacity) {
    if (!triangles || !nodes_out || n_triangles <= 0) return fail(RT_ERR_INVALID, "rt_build_bvh: null array or n_triangles <= 0");
    return guarded("rt_build_bvh", [&]() -> int {
        std::vector<rt_bvh_node> nodes;
        const int n = build_bvh(triangles, n_triangles, nodes);
        if (n > capacity) return fail(RT_ERR_CAPACITY, "rt_build_bvh: need " + std::to_string(n) + " nodes");
        std::memcpy(nodes_out, nodes.data(), (size_t)n * sizeof(rt_bvh_node));
        return n;
    });
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
    (void)hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device_id);
    if (c->n_cu <= 0) c->n_cu = 256;
    hipError_t e = hipMalloc((void**)&c->d_counters, RT_CNT_WORDS * sizeof(unsigned long long));
    if (e != hipSuccess) { delete c; return fail(RT_ERR_DEVICE, std::string("hipMalloc counters: ") + hipGetErrorString(e)); }
    (void)hipMemset(c->d_counters, 0, RT_CNT_WORDS * sizeof(unsigned long long));
    if (device_id < RT_MAX_DEVICES && !g_static_slots[device_id]) {
        // this device's copy of the static slot array (the device is current: hipSetDevice above)
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            (void)hipFree(c->d_counters);
            delete c;
            return fail(RT_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        }
        g_static_slots[device_id] = (float*)p;
    }
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_context* ctx) {
    if (!ctx) return RT_OK;
    if (ctx->job.joinable()) ctx->job.join();
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_scene(ctx);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    if (ctx->d_tune) (void)hipFree(ctx->d_tune);
    if (ctx->d_band_nodes) (void)hipFree(ctx->d_band_nodes);
    if (ctx->d_band_roots) (void)hipFree(ctx->d_band_roots);
    free_pools(ctx);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    delete ctx;
    return RT_OK;
}

int rt_set_option(rt_context* ctx, int option, int value) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_set_option: null context");
    switch (option) {
    case RT_OPT_COUNTERS: ctx->opt_counters = value ? 1 : 0; return RT_OK;
    case RT_OPT_XCD_REMAP: ctx->opt_xcd = value ? 1 : 0; return RT_OK;
    case RT_OPT_TIMING: ctx->opt_timing = value ? 1 : 0; return RT_OK;
    case RT_OPT_GPU_BVH: ctx->opt_gpu_bvh = value ? 1 : 0; return RT_OK;
    case RT_OPT_HOT_TUNE:
        if (value < 0 || value > 2) return fail(RT_ERR_INVALID, "rt_set_option: hot-node tuning must be 0, 1 or 2");
        if (value != ctx->opt_hot_tune) { ctx->hot_dirty = true; ctx->n_bands = 0; }
        ctx->opt_hot_tune = value; return RT_OK;
    case RT_OPT_SHARE_PRIMARY: ctx->opt_share_primary = value ? 1 : 0; return RT_OK;
    case RT_OPT_SHARE_EQUAL: ctx->opt_share_equal = value ? 1 : 0; return RT_OK;
    case RT_OPT_MODE:
        if (value != 0 && value != 1) return fail(RT_ERR_INVALID, "rt_set_option: mode must be 0 (megakernel) or 1 (wavefront)");
        ctx->opt_mode = value; return RT_OK;
    case RT_OPT_WF_CAPACITY_LOG2:
        if (value < 10 || value > 28) return fail(RT_ERR_INVALID, "rt_set_option: wavefront capacity must be 2^10..2^28 paths");
        ctx->opt_wf_log2 = value; return RT_OK;
    case RT_OPT_WF_SORT: ctx->opt_wf_sort = value ? 1 : 0; return RT_OK;
    case RT_OPT_WF_VOTE:
        if (value < 0 || value > 1024) return fail(RT_ERR_INVALID, "rt_set_option: vote weight must be 0..1024");
        ctx->opt_wf_vote = value; return RT_OK;
    case RT_OPT_WF_REFILL:
        if (value < 1 || value > 64) return fail(RT_ERR_INVALID, "rt_set_option: refill threshold must be 1..64 lanes");
        ctx->opt_wf_refill = value; return RT_OK;
    case RT_OPT_WF_LANES:
        if (value < 1 || value > 2) return fail(RT_ERR_INVALID, "rt_set_option: pipelines per pass must be 1 or 2");
        ctx->opt_wf_lanes = value; return RT_OK;
    case RT_OPT_WF_BLOCK:
        if (value < 64 || value > RT_EXTEND_BLOCK_MAX || value % 64) return fail(RT_ERR_INVALID, "rt_set_option: closest-hit workgroup must be a multiple of 64 up to 1024 lanes");
        ctx->opt_wf_block = value; return RT_OK;
    case RT_OPT_HOT_NODES:
        if (value < 0 || value > 512) return fail(RT_ERR_INVALID, "rt_set_option: hot nodes must be 0..512");
        ctx->opt_hot_nodes = value; return RT_OK;
    case RT_OPT_WF_RUN:
        if (value < 64 || value > (1 << 20)) return fail(RT_ERR_INVALID, "rt_set_option: closest-hit chunk must be 64..2^20 entries");
        ctx->opt_wf_run = value; return RT_OK;
    case RT_OPT_WF_BURST:
        if (value < 0 || value > (1 << 20)) return fail(RT_ERR_INVALID, "rt_set_option: node burst must be >= 0");
        ctx->opt_wf_burst = value; return RT_OK;
    case RT_OPT_LDS_SLOTS:
        if (value < 1 || value > 32) return fail(RT_ERR_INVALID, "rt_set_option: LDS stack slots must be 1..32");
        ctx->opt_lds_slots = value; return RT_OK;
    case RT_OPT_CHUNK:
        if (value < 1 || value > 65536) return fail(RT_ERR_INVALID, "rt_set_option: chunk must be 1..65536");
        ctx->opt_chunk = value; return RT_OK;
    case RT_OPT_PASS:
        if (value < 1 || value > 65536) return fail(RT_ERR_INVALID, "rt_set_option: pass must be 1..65536");
        ctx->opt_pass = value; return RT_OK;
    case RT_OPT_GROUP_SAMPLES: ctx->opt_group_samples = value ? 1 : 0; return RT_OK;
    case RT_OPT_SLOT_RING_MB:
        if (value < 0 || value > (1 << 20)) return fail(RT_ERR_INVALID, "rt_set_option: slot ring must be 0..2^20 MiB");
        ctx->slots_ring_bytes = (size_t)value << 20; return RT_OK;
    default: return fail(RT_ERR_INVALID, "rt_set_option: unknown option");
    }
}

static int upload_scene_body(rt_context* ctx, const rt_object* objects, int n_objects, const rt_material* materials, int n_materials,
                             const rt_texture* textures, int n_textures);

int rt_upload_scene(rt_context* ctx, const rt_object* objects, int n_objects, const rt_material* materials, int n_materials,
                    const rt_texture* textures, int n_textures) {
    return guarded("rt_upload_scene", [&]() -> int { return upload_scene_body(ctx, objects, n_objects, materials, n_materials, textures, n_textures); });
}

static int upload_scene_body(rt_context* ctx, const rt_object* objects, int n_objects, const rt_material* materials, int n_materials,
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
    uint64_t bytes = 0;
    int rc;
    std::string err;
    if (ctx->opt_gpu_bvh) {
        // the tree is built on the device from the triangles alone (rt_build.hip); the callers' node arrays are not read
        HIP_TRY(hipDeviceSynchronize());
        free_scene(ctx);
        DeviceScene ds;
        if (build_scene_on_device(objects, n_objects, ds, err) != 0) return fail(RT_ERR_DEVICE, "rt_upload_scene: device build: " + err);
        ctx->d_nodes = ds.nodes4; ctx->d_isect = ds.isect; ctx->d_shade = ds.shade; ctx->d_origin = ds.origin;
        ctx->dev.root = ds.root;
        ctx->dev.last_tri = ds.last_tri;
        ctx->stack_slots = ds.stack_need > 0 ? ds.stack_need : 1;
        ctx->stats[0] = ds.n_triangles;
        ctx->stats[1] = ds.n_nodes4;
        ctx->stats[5] = (uint64_t)(ds.build_ms * 1000.0f);
        ctx->stats[6] = (uint64_t)(ds.copy_ms * 1000.0f);
        ctx->stats[7] = (uint64_t)ds.binary_depth;
        bytes = ds.bytes;
    } else {
        for (int i = 0; i < n_objects; i++)
            if (!objects[i].bvh || objects[i].n_bvh_nodes <= 0) return fail(RT_ERR_INVALID, "rt_upload_scene: object without a node array (set RT_OPT_GPU_BVH to have the tree built on the device)");
        PackedScene ps;
        if (pack_scene(objects, n_objects, ps, err) != 0) return fail(RT_ERR_INVALID, "rt_upload_scene: " + err);
        HIP_TRY(hipDeviceSynchronize());
        free_scene(ctx);
        if ((rc = upload_vec(ps.nodes4, &ctx->d_nodes, bytes))) return rc;
        const size_t n_packed_tris = ps.isect.size();
        ps.isect.push_back(TriIsect{});          // one spare record behind the last triangle: no kernel reads it, it only keeps a fetch that runs past a final one-triangle leaf inside the allocation
        if ((rc = upload_vec(ps.isect, &ctx->d_isect, bytes))) return rc;
        if ((rc = upload_vec(ps.shade, &ctx->d_shade, bytes))) return rc;
        if ((rc = upload_vec(ps.origin, &ctx->d_origin, bytes))) return rc;
        ctx->dev.root = ps.root;
        ctx->dev.last_tri = ps.last_tri;
        ctx->stack_slots = ps.max_depth > 0 ? ps.max_depth : 1;
        ctx->stats[0] = n_packed_tris;
        ctx->stats[1] = ps.nodes4.size();
        ctx->stats[5] = ctx->stats[6] = ctx->stats[7] = 0;
    }
    if ((rc = upload_vec(mats, &ctx->d_mats, bytes))) return rc;
    if ((rc = upload_vec(tex, &ctx->d_tex, bytes))) return rc;
    if ((rc = upload_vec(texels, &ctx->d_texels, bytes))) return rc;
    ctx->dev.nodes = (const Node4*)ctx->d_nodes;
    ctx->dev.isect = (const TriIsect*)ctx->d_isect;
    ctx->dev.shade = (const TriShade*)ctx->d_shade;
    ctx->dev.origin = (const TriOrigin*)ctx->d_origin;
    ctx->dev.mats = (const MatDev*)ctx->d_mats;
    ctx->dev.tex = (const TexDev*)ctx->d_tex;
    ctx->dev.texels = (const uint32_t*)ctx->d_texels;
    ctx->dev.n_materials = n_materials;
    ctx->stats[2] = (uint64_t)ctx->stack_slots;
    ctx->stats[3] = bytes;
    ctx->stats[4] = (uint64_t)n_objects;
    ctx->have_scene = true;
    ctx->hot_dirty = true;
    ctx->n_bands = 0;
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
    const CameraFrame frame = camera_frame(width, height, camera_pos, camera_dir, camera_up, fov_radians);
    // the node numbering was tuned for a view: only another view asks for another calibration pass (a worker sets the
    // same camera again before every frame)
    if (!ctx->have_camera || std::memcmp(&frame, &ctx->cam, sizeof frame) != 0) { ctx->hot_dirty = true; ctx->n_bands = 0; }
    ctx->cam = frame;
    ctx->width = width;
    ctx->height = height;
    ctx->have_camera = true;
    return RT_OK;
}

static int lds_slots(const rt_context* ctx) { return ctx->stack_slots < ctx->opt_lds_slots ? ctx->stack_slots : ctx->opt_lds_slots; }
static size_t stack_bytes(const rt_context* ctx) { return (size_t)lds_slots(ctx) * RT_BLOCK * sizeof(uint2); }
static bool stack_fits(const rt_context* ctx) { return ctx->stack_slots <= lds_slots(ctx) + RT_SPILL_SLOTS; }

// --- optional per-kernel timing with HIP events on the launch stream ---
static int timing_begin(rt_context* ctx, int kind, hipStream_t s) {
    if (!ctx->opt_timing) return -1;
    if (ctx->ev_used * 2 + 2 > ctx->ev_pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        ctx->ev_pool.push_back(a); ctx->ev_pool.push_back(b);
        ctx->ev_kind.push_back(kind);
    }
    const int k = (int)ctx->ev_used++;
    ctx->ev_kind[k] = kind;
    (void)hipEventRecord(ctx->ev_pool[2 * k], s);
    return k;
}
static void timing_end(rt_context* ctx, int k, hipStream_t s) {
    if (k >= 0) (void)hipEventRecord(ctx->ev_pool[2 * k + 1], s);
}
// Fold the recorded pairs into the totals; the caller has synchronised the stream.
static void timing_collect(rt_context* ctx) {
    for (size_t k = 0; k < ctx->ev_used; k++) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->ev_pool[2 * k], ctx->ev_pool[2 * k + 1]) == hipSuccess) {
            ctx->time_ms[ctx->ev_kind[k]] += ms;
            ctx->time_launches[ctx->ev_kind[k]]++;
        }
    }
    ctx->ev_used = 0;
}

#define RT_SLOTS_MAX_BYTES ((size_t)64 << 30)
#define RT_CTL_RING 4
#define RT_CTL_GROUP 4
static uint32_t magic_of(uint32_t d) { return d <= 1u ? 0xFFFFFFFFu : (uint32_t)((1ull << 32) / d); }
// How a pass's samples are cut into runs that share a primary ray, and the divisors that go with it (see PathParams).
static void set_runs(PathParams& pp, int run_len, bool group_samples) {
    pp.run_len = run_len;
    pp.run_base = pp.first_sample / run_len;
    pp.n_runs = (pp.first_sample + pp.n_samples - 1) / run_len - pp.run_base + 1;
    pp.per_tile = (uint32_t)pp.n_runs * 256u;
    pp.per_tile_magic = magic_of(pp.per_tile);
    pp.tiles_x_magic = magic_of((uint32_t)pp.tiles_x);
    pp.batch_magic = magic_of((uint32_t)pp.batch);
    pp.run_lanes = run_len < pp.n_samples ? run_len : pp.n_samples;      // no run of the pass has more samples than the pass
    pp.run_lanes_magic = magic_of((uint32_t)pp.run_lanes);
    // one entry per sample: the samples of a batch side by side, in groups of up to 16 -- a power of two that divides the
    // batch and the pass's first sample, so that no group straddles two batches (its samples share one primary ray)
    pp.group_log2 = 0;
    if (run_len == 1 && group_samples)
        while (pp.group_log2 < 4u && pp.batch % (2 << pp.group_log2) == 0 && pp.first_sample % (2 << pp.group_log2) == 0) pp.group_log2++;
    pp.n_full_groups = (uint32_t)pp.n_runs >> pp.group_log2;
    pp.group_rem = (uint32_t)pp.n_runs - (pp.n_full_groups << pp.group_log2);
    pp.group_rem_magic = magic_of(pp.group_rem);
}

// One pass of the wavefront variant: all paths of samples [pp.first_sample, +pp.n_samples) of every pixel.  The
// samples are split between the pipelines; every path writes its own slot, so the split changes no result.
// The ordered sum of the pass's slots into `d_output` is part of the pass: with one pipeline it follows the paths tile by
// tile through the slot ring (`ring_tiles` tiles of slots at pp.slots), with two it runs when both have drained.
static int wavefront_pass(rt_context* ctx, PathParams pp, hipStream_t s, uint32_t ring_tiles, float* carry, float* d_output, int flush) {
    const int n_lanes = (ctx->opt_wf_lanes > 1 && pp.n_samples > 1) ? 2 : 1;
    const uint32_t n_tiles = (uint32_t)pp.tiles_x * (uint32_t)pp.tiles_y;
    // the sum can only follow the paths while the planner remembers the cursor of `lifetime` refills ago
    const uint32_t lifetime = (uint32_t)std::max(1, pp.max_depth);
    const bool follow = n_lanes == 1 && lifetime < RT_CURSOR_HISTORY;
    if (!follow && ring_tiles < n_tiles) return fail(RT_ERR_INVALID, "rt_raytrace: internal: a slot ring shorter than the frame needs the following sum");
    // pool size: the option's, but no larger than the pass needs (a small frame should not launch grids sized for a
    // large one, nor hold gigabytes of pools); the pools only ever grow
    const unsigned long long pass_paths = (unsigned long long)pp.tiles_x * pp.tiles_y * 256ull * (unsigned long long)pp.n_samples;
    uint32_t cap = 1u << ctx->opt_wf_log2;
    while (cap > (1u << 16) && (unsigned long long)(cap >> 1) >= pass_paths) cap >>= 1;
    cap = std::min(cap, ctx->pool_cap_limit);    // a size that failed once is not tried again on every pass
    if (ctx->pool_capacity < cap || ctx->lanes_allocated < n_lanes) {
        HIP_TRY(hipStreamSynchronize(s));
        // a device with less free memory than the default pool takes (15 GB) gets a smaller pool, not an error
        for (;;) {
            free_pools(ctx);
            hipError_t e = hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming);
            for (int l = 0; l < n_lanes && e == hipSuccess; l++) {
                rt_context::WfLane& L = ctx->lanes[l];
                for (int k = 0; k < 2 && e == hipSuccess; k++) {
                    e = hipMalloc((void**)&L.pools.p0[k], (size_t)cap * sizeof(float4));
                    if (e == hipSuccess) e = hipMalloc((void**)&L.pools.p1[k], (size_t)cap * sizeof(float4));
                    if (e == hipSuccess) e = hipMalloc((void**)&L.pools.p2[k], (size_t)cap * sizeof(float4));
                }
                if (e == hipSuccess) e = hipMalloc((void**)&L.pools.hit, (size_t)cap * sizeof(float4));
                if (e == hipSuccess) e = hipMalloc((void**)&L.d_ctl, sizeof(WfCtl));
                if (e == hipSuccess) e = hipHostMalloc((void**)&L.h_ctl, RT_CTL_RING * sizeof(WfCtl), hipHostMallocDefault);
                for (int k = 0; k < RT_CTL_RING && e == hipSuccess; k++) e = hipEventCreateWithFlags(&L.ctl_ev[k], hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&L.done, hipEventDisableTiming);
                if (e == hipSuccess) e = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
            }
            if (e == hipSuccess) break;
            (void)hipGetLastError();
            free_pools(ctx);
            if (e != hipErrorOutOfMemory || cap <= (1u << 16)) return fail(RT_ERR_DEVICE, std::string("rt_raytrace: path pools: ") + hipGetErrorString(e));
            cap >>= 1;
            ctx->pool_cap_limit = cap;
        }
        ctx->lanes_allocated = n_lanes;
        ctx->pool_capacity = cap;
    }
    cap = std::min(cap, ctx->pool_capacity);     // what this pass uses of them (less than asked for if memory was short)
    const int slots_lds = lds_slots(ctx);
    const uint32_t rpb = (uint32_t)ctx->opt_wf_run;
    const uint32_t eb = (uint32_t)ctx->opt_wf_block;
    const uint32_t n_hot = (uint32_t)std::min<uint64_t>((uint64_t)ctx->opt_hot_nodes, ctx->stats[1]);
    // regions of the pool: as many as RT_REGIONS_MAX, each a whole number of shading workgroups
    uint32_t regions = RT_REGIONS_MAX;
    while (regions > 1 && cap / regions < RT_SHADE_BLOCK) regions >>= 1;
    uint32_t log2_region_cap = 0;
    while ((1u << log2_region_cap) < cap / regions) log2_region_cap++;
    const size_t elds = (size_t)slots_lds * eb * sizeof(uint2) + (size_t)n_hot * RT_HOT_STRIDE + (((size_t)regions + 1) * sizeof(uint32_t) + 15) / 16 * 16;
    if (elds > ctx->extend_lds_limit) {               // more than the default 64 KB of dynamic LDS has to be asked for once
        if (elds > 160u * 1024u) return fail(RT_ERR_INVALID, "rt_raytrace: stack slots and hot nodes need more than 160 KB of LDS per workgroup");
        HIP_TRY(hipFuncSetAttribute((const void*)wf_extend_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)elds));
        HIP_TRY(hipFuncSetAttribute((const void*)wf_extend_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)elds));
        HIP_TRY(hipFuncSetAttribute((const void*)wf_extend_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)elds));
        HIP_TRY(hipFuncSetAttribute((const void*)wf_extend_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)elds));
        ctx->extend_lds_limit = elds;
    }
    // the pipelines start after the work already queued on the caller's stream and hand back to it at the end
    HIP_TRY(hipEventRecord(ctx->fork_ev, s));
    struct LaneRun {
        PathParams pp;
        unsigned long long total;       // pool entries the pass starts (runs of samples x pixels of whole tiles)
        uint32_t entry_yield;           // survivors a new entry can leave at most
        int cur;
        int snap_cur[RT_CTL_RING];
        unsigned iter;
        bool drained;
        bool stop_known;                // the iteration whose shading ends the pass is known ...
        unsigned stop_at;               // ... and is this one
    } run[2];
    for (int l = 0; l < n_lanes; l++) {
        rt_context::WfLane& L = ctx->lanes[l];
        LaneRun& R = run[l];
        const int first = (int)((long long)pp.n_samples * l / n_lanes), last = (int)((long long)pp.n_samples * (l + 1) / n_lanes);
        R.pp = pp;
        R.pp.first_sample = pp.first_sample + first;
        R.pp.n_samples = last - first;
        // tile layout: [tile][sample of the pipeline's share][256]; the second pipeline's slots follow the first's
        R.pp.slots = pp.slots + (size_t)first * (size_t)ring_tiles * 256u * 3;
        R.pp.ring_tiles = ring_tiles;
        R.pp.ring_magic = magic_of(ring_tiles);
        R.pp.width_magic = magic_of((uint32_t)pp.cam.width);
        // one closest-hit query per pixel and semantic batch, unless the option says one per sample or a run's survivors
        // would take more than an eighth of a region
        const uint32_t longest_run = (uint32_t)std::min(pp.batch, R.pp.n_samples);
        const int run_len = (ctx->opt_share_primary && pp.batch > 1 && longest_run <= (cap / regions) / 8u) ? pp.batch : 1;
        set_runs(R.pp, run_len, ctx->opt_group_samples != 0);
        R.entry_yield = (uint32_t)R.pp.run_lanes;      // the shading launch gives every new entry that many lanes, whatever its run is cut to
        R.total = (unsigned long long)pp.tiles_x * pp.tiles_y * 256ull * (unsigned long long)R.pp.n_runs;
        // A pass whose entries all fit the first refill needs no look at the pool's population: iteration i traces segment
        // i of every path, so the shading of iteration max_depth ends the last of them.  (An interactive frame of a few
        // samples otherwise pays up to eight empty iterations before the snapshots tell.)
        R.stop_known = R.total <= (unsigned long long)regions * ((cap / regions) / R.entry_yield) && ring_tiles >= n_tiles;
        R.stop_at = (unsigned)std::max(1, pp.max_depth);
        // (Longer passes learn the iteration of the last start from the snapshots, a group late but well before it matters.)
        R.cur = 0;
        for (int& c : R.snap_cur) c = 0;
        R.iter = 0;
        R.drained = false;
        HIP_TRY(hipStreamWaitEvent(L.stream, ctx->fork_ev, 0));
        HIP_TRY(hipMemsetAsync(L.d_ctl, 0, sizeof(WfCtl), L.stream));
    }
    const dim3 grid(std::min<uint32_t>(cap / RT_BLOCK, (uint32_t)ctx->n_cu * 64u)), block(RT_BLOCK);
    const dim3 sgrid(cap / RT_SHADE_BLOCK), sblock(RT_SHADE_BLOCK);
    const dim3 agrid(std::min<uint32_t>(n_tiles, (uint32_t)ctx->n_cu * 16u));
    // one launch fills the chip once; the waves share out the pool themselves
    const uint32_t wanted = (cap + rpb * (eb / 64) - 1) / (rpb * (eb / 64));
    const dim3 egrid(std::min<uint32_t>(wanted, (uint32_t)ctx->n_cu * (2048u / eb))), eblock(eb);
    for (int live = n_lanes; live > 0;) {
        for (int l = 0; l < n_lanes; l++) {
            rt_context::WfLane& L = ctx->lanes[l];
            LaneRun& R = run[l];
            if (R.drained) continue;
            hipStream_t ls = L.stream;
            const int nxt = R.cur ^ 1;
            if (R.iter > 0) {
                const int tk = timing_begin(ctx, 1, ls);
                if (ctx->opt_counters) hipLaunchKernelGGL(wf_shade_kernel<true>, sgrid, sblock, 0, ls, ctx->dev, L.d_ctl, L.pools, R.cur, regions, log2_region_cap, R.pp, ctx->opt_share_equal);
                else hipLaunchKernelGGL(wf_shade_kernel<false>, sgrid, sblock, 0, ls, ctx->dev, L.d_ctl, L.pools, R.cur, regions, log2_region_cap, R.pp, ctx->opt_share_equal);
                timing_end(ctx, tk, ls);
                // iteration i traces segment i - j of the paths started in iteration j, so the shading of iteration
                // (last start) + max_depth ends the pass's last paths
                if (R.stop_known && R.iter >= R.stop_at) {
                    HIP_TRY(hipGetLastError());
                    R.drained = true; live--;
                    continue;
                }
            }
            hipLaunchKernelGGL(wf_plan_kernel, dim3(1), dim3(RT_REGIONS_MAX), 0, ls, L.d_ctl, nxt, R.total, regions, cap / regions, R.pp.per_tile, R.entry_yield,
                               follow ? ring_tiles : 0u, lifetime, (uint32_t)R.pp.tiles_x, (uint32_t)R.pp.tiles_y, ctx->n_bands, ctx->d_counters);
            // the tiles whose last path has ended are summed now, before any later tile's paths write to their places in the ring
            // (On a stream of its own beside the closest-hit launch the sum cost more than it hid: it pushes 30 GB per pass
            // through the L2 the walk lives in -- closest-hit launches +4 %.)
            if (follow && R.iter >= lifetime)
                hipLaunchKernelGGL(accumulate_tiles_kernel, agrid, dim3(RT_BLOCK), 0, ls, (const WfCtl*)L.d_ctl, 0, n_tiles, ring_tiles, R.pp.tiles_x, R.pp.cam.width,
                                   R.pp.cam.height, R.pp.first_sample, R.pp.n_samples, R.pp.batch, flush, (const float*)R.pp.slots, carry, d_output);
            hipLaunchKernelGGL(wf_generate_kernel, grid, block, 0, ls, (const WfCtl*)L.d_ctl, L.pools, nxt, regions, log2_region_cap, R.pp);
            const int tke = timing_begin(ctx, 0, ls);
            {
                // four builds of one kernel: with or without the counters, with the per-trip vote or the fixed node burst
                auto kernel = ctx->opt_counters ? (ctx->opt_wf_vote ? wf_extend_kernel<true, true> : wf_extend_kernel<true, false>)
                                                : (ctx->opt_wf_vote ? wf_extend_kernel<false, true> : wf_extend_kernel<false, false>);
                hipLaunchKernelGGL(kernel, egrid, eblock, elds, ls, ctx->dev, L.d_ctl, L.pools, nxt,
                                   slots_lds, rpb, ctx->opt_wf_burst, n_hot, ctx->opt_wf_vote, ctx->opt_wf_refill, R.pp.cam.pos[0], R.pp.cam.pos[1], R.pp.cam.pos[2],
                                   regions, log2_region_cap, ctx->d_counters, (const Node4*)ctx->d_band_nodes, (const uint32_t*)ctx->d_band_roots,
                                   ctx->n_bands ? ctx->band_records : 0u);
            }
            timing_end(ctx, tke, ls);
            HIP_TRY(hipGetLastError());
            R.cur = nxt;
            // The host does not know the pool's population.  After every group of iterations it queues a snapshot
            // of the control block, and it reads the snapshot of the group BEFORE, so the stream always holds
            // a group of work while the host waits; the price is one group of empty launches at the very end.
            if ((R.iter % RT_CTL_GROUP) == RT_CTL_GROUP - 1) {
                const unsigned g = R.iter / RT_CTL_GROUP;
                HIP_TRY(hipMemcpyAsync(&L.h_ctl[g % RT_CTL_RING], L.d_ctl, RT_CTL_HEAD_BYTES, hipMemcpyDeviceToHost, ls));
                HIP_TRY(hipEventRecord(L.ctl_ev[g % RT_CTL_RING], ls));
                R.snap_cur[g % RT_CTL_RING] = R.cur;
                if (g > 0) {
                    const unsigned q = (g - 1) % RT_CTL_RING;
                    HIP_TRY(hipEventSynchronize(L.ctl_ev[q]));
                    if (L.h_ctl[q].cursor >= R.total && L.h_ctl[q].count_total[R.snap_cur[q]] == 0) { R.drained = true; live--; }
                    else if (L.h_ctl[q].cursor >= R.total && !R.stop_known) {
                        R.stop_known = true;
                        R.stop_at = L.h_ctl[q].last_gen_iter + (unsigned)std::max(1, pp.max_depth);
                    }
                }
            }
            if (++R.iter > (1u << 26)) return fail(RT_ERR_DEVICE, "rt_raytrace: wavefront pass did not drain");
        }
    }
    for (int l = 0; l < n_lanes; l++) {
        HIP_TRY(hipEventRecord(ctx->lanes[l].done, ctx->lanes[l].stream));
        HIP_TRY(hipStreamWaitEvent(s, ctx->lanes[l].done, 0));
    }
    // what the following sum has not taken (everything, without it), on the caller's stream: the pipelines' shares in sample
    // order, a batch that straddles two shares carried from one to the next like a batch that straddles two passes
    for (int l = 0; l < n_lanes; l++) {
        const LaneRun& R = run[l];
        hipLaunchKernelGGL(accumulate_tiles_kernel, agrid, dim3(RT_BLOCK), 0, s, (const WfCtl*)ctx->lanes[l].d_ctl, 1, n_tiles, ring_tiles, R.pp.tiles_x, R.pp.cam.width,
                           R.pp.cam.height, R.pp.first_sample, R.pp.n_samples, R.pp.batch, (flush && l == n_lanes - 1) ? 1 : 0, (const float*)R.pp.slots, carry, d_output);
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

// Which node records the closest-hit kernel should keep in LDS depends on the scene AND the camera: the records rays
// enter most.  The static numbering (largest boxes first) is a guess; this measures.  One sample per pixel of a reduced
// frame is traced with the counting build of the megakernel -- same rays, same walk as the frame's -- into a histogram of visits over
// the first RT_HIST_NODES records, and those records are renumbered by falling count (ties: old order).  Node numbers
// are only addresses: no result changes.  Costs about ten milliseconds and a stream
// synchronisation, so it runs once per scene and camera, and only for calls of at least RT_TUNE_MIN_SPP samples.
#define RT_TUNE_MIN_SPP 64
// RT_OPT_HOT_TUNE = 2 goes one step further.  A pass starts its paths tile row after tile row, so at any moment the pools
// hold paths of a band of the picture a few tile rows tall, and the records those paths enter most are not the whole
// view's: the calibration counts visits per band (up to RT_BANDS_MAX equal shares of the tile rows), and for every band a
// copy of the first records is numbered by the visits of that band and its two neighbours.  The refill planner names
// the band in flight in the control block and the closest-hit launch reads the first records -- the ones it copies to LDS
// first of all -- from that band's copy.  Measured on formula1: 51 % of node visits served from LDS with one numbering,
// 55-58 % with the band's own.
#define RT_BANDS_MAX 64u
static int tune_hot_nodes(rt_context* ctx, PathParams pp, hipStream_t s) {
    const uint32_t k = (uint32_t)std::min<uint64_t>(RT_HIST_NODES, ctx->stats[1]);
    ctx->n_bands = 0;
    if (k < 2) return RT_OK;
    pp.n_samples = 1;
    pp.chunk = 1;
    pp.n_chunks = 1;
    // the same view at a quarter of the resolution each way: a sixteenth of the paths tells the visit counts of a few
    // thousand records apart just as well (the counting megakernel is slow: a full-resolution sample cost 190 ms)
    pp.cam.width = std::max(16, pp.cam.width / 4);
    pp.cam.height = std::max(16, pp.cam.height / 4);
    pp.n_pixel = pp.cam.width * pp.cam.height;
    pp.tiles_x = (pp.cam.width + 15) / 16;
    pp.tiles_y = (pp.cam.height + 15) / 16;
    // a band's numbering is as good as its share of the calibration's paths: at least 2048 of them per band, or fewer bands
    uint32_t bands = ctx->opt_hot_tune >= 2 ? std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(RT_BANDS_MAX, (uint32_t)pp.tiles_y), (uint32_t)pp.n_pixel / 2048u)) : 1u;
    const size_t hist_bytes = (size_t)RT_BANDS_MAX * RT_HIST_NODES * sizeof(uint32_t), cnt_bytes = RT_CNT_WORDS * sizeof(unsigned long long);
    const size_t perm_at = hist_bytes + cnt_bytes, copy_at = perm_at + hist_bytes;       // the permutations take as much room as the counts
    if (!ctx->d_tune) HIP_TRY(hipMalloc(&ctx->d_tune, copy_at + RT_HIST_NODES * sizeof(Node4)));
    char* base = (char*)ctx->d_tune;
    uint32_t* d_hist = (uint32_t*)base;
    uint32_t* d_perm = (uint32_t*)(base + perm_at);
    Node4* d_copy = (Node4*)(base + copy_at);
    HIP_TRY(hipMemsetAsync(base, 0, hist_bytes + cnt_bytes, s));
    pp.counters = (unsigned long long*)(base + hist_bytes);      // the calibration's rays are not the frame's: counted apart and dropped
    pp.hist = d_hist;
    pp.hist_bands = bands > 1u ? (int)bands : 0;
    const long long nblocks = (long long)pp.tiles_x * pp.tiles_y;
    hipLaunchKernelGGL(path_kernel<true>, dim3((unsigned)nblocks), dim3(RT_BLOCK), stack_bytes(ctx), s, ctx->dev, pp);
    HIP_TRY(hipGetLastError());
    std::vector<uint32_t> counts((size_t)bands * RT_HIST_NODES);
    HIP_TRY(hipMemcpyAsync(counts.data(), d_hist, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // the whole view's numbering: the array itself, which every other kernel reads
    std::vector<uint64_t> hist(k, 0);
    for (uint32_t b = 0; b < bands; b++)
        for (uint32_t i = 0; i < k; i++) hist[i] += counts[(size_t)b * RT_HIST_NODES + i];
    std::vector<uint32_t> order(k), perm(k);
    for (uint32_t i = 0; i < k; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&hist](uint32_t a, uint32_t b) { return hist[a] > hist[b]; });
    bool moved = false;
    for (uint32_t at = 0; at < k; at++) { perm[order[at]] = at; moved = moved || order[at] != at; }
    if (moved) {
        HIP_TRY(hipMemcpyAsync(d_perm, perm.data(), k * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(permute_nodes_kernel, dim3((k + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, s, (const Node4*)ctx->d_nodes, k, (const uint32_t*)d_perm, d_copy);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ctx->d_nodes, d_copy, (size_t)k * sizeof(Node4), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));                         // d_perm is written again below
        uint32_t& root = ctx->dev.root.ref;
        if (root != REF_NONE && !(root & REF_LEAF) && root < k) root = perm[root];
    }
    ctx->hot_dirty = false;
    if (bands < 2u) return RT_OK;
    // one copy per band, numbered by the visits of the band and its neighbours (counts indexed by the numbers the array has now)
    if (!ctx->d_band_nodes) HIP_TRY(hipMalloc((void**)&ctx->d_band_nodes, (size_t)RT_BANDS_MAX * RT_HIST_NODES * sizeof(Node4)));
    if (!ctx->d_band_roots) HIP_TRY(hipMalloc((void**)&ctx->d_band_roots, RT_BANDS_MAX * sizeof(uint32_t)));
    const int window = 1;          // a band alone: -0.5 points of LDS-served visits; two neighbours each side: -0.6
    std::vector<uint32_t> perms((size_t)bands * RT_HIST_NODES, 0u), roots(RT_BANDS_MAX, REF_NONE);
    std::vector<uint64_t> mine(k);
    const uint32_t root_now = ctx->dev.root.ref;
    for (uint32_t b = 0; b < bands; b++) {
        std::fill(mine.begin(), mine.end(), 0ull);
        for (int nb = (int)b - window; nb <= (int)b + window; nb++) {
            if (nb < 0 || nb >= (int)bands) continue;
            const uint32_t* c = counts.data() + (size_t)nb * RT_HIST_NODES;
            for (uint32_t i = 0; i < k; i++) mine[perm[i]] += (nb == (int)b ? 2ull : 1ull) * c[i];     // perm: old number -> number in the array now
        }
        for (uint32_t i = 0; i < k; i++) order[i] = i;
        // ties (records the band never entered) keep the whole view's order
        std::stable_sort(order.begin(), order.end(), [&mine](uint32_t a, uint32_t c) { return mine[a] > mine[c]; });
        uint32_t* pb = perms.data() + (size_t)b * RT_HIST_NODES;
        for (uint32_t at = 0; at < k; at++) pb[order[at]] = at;
        roots[b] = (root_now != REF_NONE && !(root_now & REF_LEAF) && root_now < k) ? pb[root_now] : root_now;
    }
    HIP_TRY(hipMemcpyAsync(d_perm, perms.data(), perms.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->d_band_roots, roots.data(), RT_BANDS_MAX * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(permute_bands_kernel, dim3((k + RT_BLOCK - 1) / RT_BLOCK, bands), dim3(RT_BLOCK), 0, s, (const Node4*)ctx->d_nodes, k, (const uint32_t*)d_perm, ctx->d_band_nodes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));                             // `perms` and `roots` leave scope
    ctx->n_bands = bands;
    ctx->band_records = k;
    return RT_OK;
}

// Body of rt_raytrace and rt_raytrace_static.  `fixed_slots` == nullptr: the slot buffer belongs to the context and grows
// on demand, the context's kernel design is used.  Otherwise the slots are `fixed_bytes` of memory the caller of this
// function provides, nothing is allocated, created or freed, and every pass is one path_kernel launch.
static int raytrace_body(const char* who, rt_context* ctx, int n_pixel, int first_sample, int spp, int semantic_batch, int max_depth,
                         const float ambient[3], float russian_roulette, float* d_output, hipStream_t s, float* fixed_slots,
                         size_t fixed_bytes) {
    const std::string name(who);
    if (!ctx || !ambient || !d_output) return fail(RT_ERR_INVALID, name + ": null argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, name + ": no scene uploaded");
    if (!ctx->have_camera) return fail(RT_ERR_NO_CAMERA, name + ": rt_setup_raytrace not called");
    if (n_pixel != ctx->cam.width * ctx->cam.height) return fail(RT_ERR_INVALID, name + ": n_pixel does not match the camera frame");
    if (spp < 0 || first_sample < 0 || semantic_batch <= 0 || max_depth < 0) return fail(RT_ERR_INVALID, name + ": negative count");
    if (first_sample % semantic_batch) return fail(RT_ERR_INVALID, name + ": first_sample must be a multiple of semantic_batch");
    if ((long long)first_sample + spp > 0x7FFFFFFFLL - semantic_batch) return fail(RT_ERR_INVALID, name + ": sample index overflow");
    if (spp == 0) return RT_OK;
    if (n_pixel > (int)RT_PIXEL_MASK) return fail(RT_ERR_INVALID, name + ": more than 2^26-1 pixels");
    if (!stack_fits(ctx)) return fail(RT_ERR_INVALID, name + ": BVH deeper than the traversal stack allows");
    HIP_TRY(hipSetDevice(ctx->device));
    // Slot buffer.  Megakernel: one float3 per (sample of the pass, pixel), then one carry plane; the pass is as long as
    // the buffer allows.  Wavefront: a ring of 16x16-pixel tiles, each with a float3 per (sample of the pass, pixel of
    // the tile), then the carry plane: the pass starts its paths tile after tile and sums a tile's slots as soon as its
    // last path has ended, so the ring only has to hold the tiles that have paths in flight -- about
    // RT_SLOTS_RING_BYTES whatever the frame, and a pass can be as long as RT_OPT_PASS says.
    const bool wavefront = !fixed_slots && ctx->opt_mode == 1;
    const size_t plane = (size_t)n_pixel * 3 * sizeof(float);
    int pass = spp < ctx->opt_pass ? spp : ctx->opt_pass;
    const uint32_t n_tiles = (uint32_t)((ctx->cam.width + 15) / 16) * (uint32_t)((ctx->cam.height + 15) / 16);
    uint32_t ring_tiles = 0;
    float* slots = fixed_slots;
    float* carry = nullptr;
    if (wavefront) {
        const bool one_pipeline = !(ctx->opt_wf_lanes > 1 && pass > 1) && std::max(1, max_depth) < (int)RT_CURSOR_HISTORY;
        auto ring_for = [&](int pass_len) {
            const size_t tile_bytes = (size_t)pass_len * 256u * 3 * sizeof(float);
            const size_t want = one_pipeline ? std::max<size_t>(1, ctx->slots_ring_bytes / tile_bytes) : (size_t)n_tiles;
            return (uint32_t)std::min<size_t>(n_tiles, want);
        };
        ring_tiles = ring_for(pass);
        size_t need = (size_t)ring_tiles * (size_t)pass * 256u * 3 * sizeof(float) + plane;
        if (need > ctx->slots_bytes) {
            HIP_TRY(hipStreamSynchronize(s));
            if (ctx->d_slots) { HIP_TRY(hipFree(ctx->d_slots)); ctx->d_slots = nullptr; ctx->slots_bytes = 0; }
            // a device with less free memory than the defaults assume gets a shorter ring (or shorter passes), not an error
            hipError_t e = hipMalloc((void**)&ctx->d_slots, need);
            while (e == hipErrorOutOfMemory && (ring_tiles > 1 || pass > 1)) {
                (void)hipGetLastError();
                if (one_pipeline && ring_tiles > 64) ring_tiles = (ring_tiles + 1) / 2;
                else { pass = (pass + 1) / 2; ring_tiles = std::min(ring_tiles, ring_for(pass)); }
                need = (size_t)ring_tiles * (size_t)pass * 256u * 3 * sizeof(float) + plane;
                e = hipMalloc((void**)&ctx->d_slots, need);
            }
            if (e != hipSuccess) { ctx->d_slots = nullptr; return fail(RT_ERR_DEVICE, name + ": hipMalloc of the sample slots: " + hipGetErrorString(e)); }
            ctx->slots_bytes = need;
        }
        slots = ctx->d_slots;
        carry = slots + (size_t)ring_tiles * (size_t)pass * 256u * 3;
    } else {
        const size_t fit = (fixed_slots ? fixed_bytes : RT_SLOTS_MAX_BYTES) / plane;
        if (fixed_slots && fit < 2)
            return fail(RT_ERR_CAPACITY, name + ": a frame of " + std::to_string(n_pixel) + " pixels needs " + std::to_string(2 * plane) +
                                         " bytes of sample slots, the static array holds " + std::to_string(fixed_bytes));
        if (fit >= 2 && (size_t)pass > fit - 1) pass = (int)(fit - 1);
        if (!fixed_slots) {
            size_t need = ((size_t)pass + 1) * plane;
            if (need > ctx->slots_bytes) {
                HIP_TRY(hipStreamSynchronize(s));
                if (ctx->d_slots) { HIP_TRY(hipFree(ctx->d_slots)); ctx->d_slots = nullptr; ctx->slots_bytes = 0; }
                hipError_t e = hipMalloc((void**)&ctx->d_slots, need);
                while (e == hipErrorOutOfMemory && pass > 1) {
                    (void)hipGetLastError();
                    pass = (pass + 1) / 2;
                    need = ((size_t)pass + 1) * plane;
                    e = hipMalloc((void**)&ctx->d_slots, need);
                }
                if (e != hipSuccess) { ctx->d_slots = nullptr; return fail(RT_ERR_DEVICE, name + ": hipMalloc of the sample slots: " + hipGetErrorString(e)); }
                ctx->slots_bytes = need;
            }
            slots = ctx->d_slots;
        }
        carry = slots + (size_t)pass * (size_t)n_pixel * 3;
    }
    PathParams pp;
    pp.cam = ctx->cam;
    pp.n_pixel = n_pixel;
    pp.batch = semantic_batch;
    pp.max_depth = max_depth;
    pp.ambient[0] = ambient[0]; pp.ambient[1] = ambient[1]; pp.ambient[2] = ambient[2];
    pp.rr = russian_roulette;
    pp.key0 = (uint32_t)ctx->seed; pp.key1 = (uint32_t)(ctx->seed >> 32);
    pp.slots = slots;
    pp.counters = ctx->d_counters;
    pp.tiles_x = (ctx->cam.width + 15) / 16;
    pp.tiles_y = (ctx->cam.height + 15) / 16;
    pp.xcd_remap = ctx->opt_xcd;
    pp.chunk = ctx->opt_chunk;
    pp.lds_slots = lds_slots(ctx);
    pp.sort_rays = ctx->opt_wf_sort;
    pp.hist = nullptr; pp.hist_bands = 0;
    pp.per_tile = pp.per_tile_magic = pp.tiles_x_magic = pp.batch_magic = 0;     // set per pass by the wavefront variant
    pp.run_len = 1; pp.run_base = 0; pp.n_runs = 0; pp.run_lanes = 1; pp.run_lanes_magic = 0xFFFFFFFFu;
    pp.ring_tiles = 0; pp.ring_magic = 0; pp.width_magic = 0;                   // the linear slot layout, unless the wavefront pass says otherwise
    pp.group_log2 = 0; pp.n_full_groups = 0; pp.group_rem = 0; pp.group_rem_magic = 0;
    if (wavefront && ctx->opt_hot_tune && ctx->hot_dirty && spp >= RT_TUNE_MIN_SPP && ctx->opt_hot_nodes > 0) {
        pp.first_sample = first_sample;
        const int rc = tune_hot_nodes(ctx, pp, s);
        if (rc) return rc;
    }
    for (int done = 0; done < spp; done += pass) {
        pp.first_sample = first_sample + done;
        pp.n_samples = spp - done < pass ? spp - done : pass;
        pp.n_chunks = (pp.n_samples + pp.chunk - 1) / pp.chunk;
        const long long nblocks = (long long)pp.tiles_x * pp.tiles_y * pp.n_chunks;
        if (nblocks > 0x7FFFFFFFLL) return fail(RT_ERR_INVALID, name + ": grid too large");
        const int last = done + pass >= spp ? 1 : 0;
        if (wavefront) {
            const int rc = wavefront_pass(ctx, pp, s, ring_tiles, carry, d_output, last);
            if (rc) return rc;
            continue;                                   // the pass summed its own slots
        } else {
            const int tk = fixed_slots ? -1 : timing_begin(ctx, 2, s);     // timing creates events on demand
            if (ctx->opt_counters) hipLaunchKernelGGL(path_kernel<true>, dim3((unsigned)nblocks), dim3(RT_BLOCK), stack_bytes(ctx), s, ctx->dev, pp);
            else hipLaunchKernelGGL(path_kernel<false>, dim3((unsigned)nblocks), dim3(RT_BLOCK), stack_bytes(ctx), s, ctx->dev, pp);
            timing_end(ctx, tk, s);
        }
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(accumulate_kernel, dim3((n_pixel + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, s, n_pixel, pp.first_sample,
                           pp.n_samples, semantic_batch, last, (const float*)slots, carry, d_output);
        HIP_TRY(hipGetLastError());
    }
    return RT_OK;
}

int rt_raytrace(rt_context* ctx, int n_pixel, int first_sample, int spp, int semantic_batch, int max_depth, const float ambient[3],
                float russian_roulette, float* d_output, void* stream) {
    return guarded("rt_raytrace", [&]() -> int {
        return raytrace_body("rt_raytrace", ctx, n_pixel, first_sample, spp, semantic_batch, max_depth, ambient, russian_roulette, d_output,
                             (hipStream_t)stream, nullptr, 0);
    });
}

int rt_raytrace_start(rt_context* ctx, int n_pixel, int first_sample, int spp, int semantic_batch, int max_depth, const float ambient[3],
                      float russian_roulette, float* d_output, void* stream) {
    if (!ctx || !ambient) return fail(RT_ERR_INVALID, "rt_raytrace_start: null argument");
    if (ctx->job.joinable()) return fail(RT_ERR_INVALID, "rt_raytrace_start: the context is still driving a call (rt_raytrace_wait first)");
    const float amb[3] = {ambient[0], ambient[1], ambient[2]};
    ctx->job_status = RT_OK;
    ctx->job_error.clear();
    try {
        ctx->job = std::thread([=]() {
            const int rc = guarded("rt_raytrace_start", [&]() -> int {
                return raytrace_body("rt_raytrace_start", ctx, n_pixel, first_sample, spp, semantic_batch, max_depth, amb, russian_roulette,
                                     d_output, (hipStream_t)stream, nullptr, 0);
            });
            ctx->job_status = rc;
            if (rc != RT_OK) ctx->job_error = g_last_error;       // the error text is per thread: hand it to the one that waits
        });
    } catch (const std::exception& e) {
        return fail(RT_ERR_HOST, std::string("rt_raytrace_start: cannot start a thread: ") + e.what());
    }
    return RT_OK;
}

int rt_raytrace_wait(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_raytrace_wait: null argument");
    if (!ctx->job.joinable()) return RT_OK;                       // nothing outstanding
    ctx->job.join();
    if (ctx->job_status != RT_OK) return fail(ctx->job_status, ctx->job_error);
    return RT_OK;
}

int rt_raytrace_static(rt_context* ctx, int n_pixel, int first_sample, int spp, int semantic_batch, int max_depth, const float ambient[3],
                       float russian_roulette, float* d_output, void* stream) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_raytrace_static: null argument");
    float* slots = ctx->device < RT_MAX_DEVICES ? g_static_slots[ctx->device] : nullptr;
    if (!slots) return fail(RT_ERR_DEVICE, "rt_raytrace_static: no static slot array on this device");
    return raytrace_body("rt_raytrace_static", ctx, n_pixel, first_sample, spp, semantic_batch, max_depth, ambient, russian_roulette, d_output,
                         (hipStream_t)stream, slots, (size_t)RT_STATIC_SCRATCH_BYTES);
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
    std::vector<uint64_t> all(RT_CNT_WORDS);
    HIP_TRY(hipMemcpy(all.data(), ctx->d_counters, RT_CNT_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (int k = 0; k < RT_CNT_COUNT; k++) out[k] = all[k];
    for (uint32_t b = 0; b < RT_CNT_SPREAD; b++) {            // the shading kernel's share of these three, kept in RT_CNT_SPREAD places
        out[RT_CNT_PATHS] += all[RT_CNT_COUNT + b * RT_CNT_SPREAD_STRIDE];
        out[RT_CNT_RAYS] += all[RT_CNT_COUNT + b * RT_CNT_SPREAD_STRIDE + 1];
        out[RT_CNT_WALKS] += all[RT_CNT_COUNT + b * RT_CNT_SPREAD_STRIDE + 2];
    }
    return RT_OK;
}

int rt_get_timing(rt_context* ctx, double out[8]) {
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_get_timing: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    timing_collect(ctx);
    for (int k = 0; k < 3; k++) { out[2 * k] = ctx->time_ms[k]; out[2 * k + 1] = (double)ctx->time_launches[k]; }
    out[6] = out[7] = 0.0;
    return RT_OK;
}

int rt_reset_counters(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_INVALID, "rt_reset_counters: null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(ctx->d_counters, 0, RT_CNT_WORDS * sizeof(uint64_t)));
    timing_collect(ctx);
    for (int k = 0; k < 3; k++) { ctx->time_ms[k] = 0.0; ctx->time_launches[k] = 0; }
    return RT_OK;
}

int rt_scene_stats(rt_context* ctx, uint64_t out[8]) {
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_scene_stats: null argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "rt_scene_stats: no scene uploaded");
    std::memcpy(out, ctx->stats, sizeof(ctx->stats));
    return RT_OK;
}

int rt_check_node_numbering(rt_context* ctx, uint64_t out[8]) {
    if (!ctx || !out) return fail(RT_ERR_INVALID, "rt_check_node_numbering: null argument");
    if (!ctx->have_scene) return fail(RT_ERR_NO_SCENE, "rt_check_node_numbering: no scene uploaded");
    return guarded("rt_check_node_numbering", [&]() -> int {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipDeviceSynchronize());
        const uint64_t n = ctx->stats[1];
        std::vector<Node4> nodes(n);
        if (n) HIP_TRY(hipMemcpy(nodes.data(), ctx->d_nodes, n * sizeof(Node4), hipMemcpyDeviceToHost));
        const uint64_t k = std::min<uint64_t>(RT_HIST_NODES, n);
        for (int i = 0; i < 8; i++) out[i] = 0;
        out[0] = n; out[1] = k;
        for (uint64_t i = 0; i < n; i++)
            for (int slot = 0; slot < 4; slot++) {
                const uint32_t r = nodes[i].ref[slot];
                if (r == REF_NONE || (r & REF_LEAF)) continue;
                if (i >= k && r < k) out[2]++;
                if (r >= n) out[3]++;
                if (r <= i) out[4]++;
            }
        return RT_OK;
    });
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
