/* rtcore.h -- C ABI of the MI355X path-tracing core (librtcore.so).
 *
 * The upstream renderer (s7a9/RayTrace) has no plugin ABI; its seam is the set
 * of C++ free functions that Worker/SceneLoader call into the .cu files.  Each
 * entry point below names the upstream declaration it replaces (file:line,
 * relative to the upstream repository root).  Only plain pointers, sizes and
 * PODs cross this boundary.  Every function returns 0 on success and a
 * negative rt_status on failure; rt_last_error() gives the message of the last
 * failure on the calling thread.  Each call selects its context's device
 * itself, so a context may be driven from any one thread at a time.
 *
 * Floating-point contract: IEEE binary32, no FMA contraction, correctly
 * rounded divide/sqrt -- identical on the host halves and in the kernels, so
 * that hit records are bit-reproducible.
 */
#ifndef RTCORE_H
#define RTCORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTCORE_ABI_VERSION 2

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument */
    RT_ERR_DEVICE = -2,       /* HIP runtime error */
    RT_ERR_NO_SCENE = -3,     /* call needs rt_upload_scene first */
    RT_ERR_NO_CAMERA = -4,    /* call needs rt_setup_raytrace first */
    RT_ERR_CAPACITY = -5,     /* caller buffer too small */
    RT_ERR_HOST = -6          /* the host side of a call failed: out of host memory, no thread to be had (no C++ exception leaves the library) */
} rt_status;

/* ---- PODs in the upstream host layout (so upstream arrays pass unchanged) ---- */

typedef struct rt_float3 { float x, y, z; } rt_float3;

/* include/primitives.cuh:22-26 -- CUDA float2 is 8-byte aligned, hence the alignment. */
typedef struct rt_vertex {
    rt_float3 position;
    rt_float3 normal;
    float texcoord[2];
} __attribute__((aligned(8))) rt_vertex;                     /* 32 bytes */

/* include/primitives.cuh:28-31 */
typedef struct rt_triangle {
    int32_t material_id;
    rt_vertex v[3];
} rt_triangle;                                               /* 104 bytes */

/* include/bvh.h:62-65 (bool is_leaf; AABB box; int left, right) */
typedef struct rt_bvh_node {
    uint8_t is_leaf;
    uint8_t pad_[3];
    rt_float3 box_min;
    rt_float3 box_max;
    int32_t left, right;
} rt_bvh_node;                                               /* 36 bytes */

/* include/renderobject.cuh:10-15, with HOST pointers */
typedef struct rt_object {
    const rt_triangle* triangles;
    const rt_bvh_node* bvh;
    int32_t n_triangles;
    int32_t n_bvh_nodes;
} rt_object;

/* include/material.cuh:10-26; the CUDA array/texture handles become an index
 * into the texture table passed to rt_upload_scene (-1 = untextured). */
enum { RT_MAT_NONE = 0, RT_MAT_LIGHT, RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_REFRACTIVE, RT_MAT_REFLECTIVE };
typedef struct rt_material {
    int32_t type;
    rt_float3 albedo;
    float optical_density;
    float metal_fuzz;
    int32_t texture;
    int32_t pad_;
} rt_material;                                               /* 32 bytes */

/* csrc/sceneloader.cpp:223-252: RGBA8, row 0 = bottom row (already flipped). */
typedef struct rt_texture {
    const uint8_t* rgba;
    int32_t width, height;
} rt_texture;

/* include/transform.cuh:10-14 */
typedef struct rt_transform {
    rt_float3 scale;
    rt_float3 translation;
    rt_float3 rotation;       /* degrees */
} rt_transform;

typedef struct rt_context rt_context;

/* counters returned by rt_get_counters */
enum {
    RT_CNT_RAYS = 0,          /* closest-hit queries as upstream runs them: one per sample and depth-loop iteration (cuda/raytrace.cu:213-219) */
    RT_CNT_BOX_TESTS,         /* AABB slab tests */
    RT_CNT_TRI_TESTS,         /* Moller-Trumbore tests */
    RT_CNT_SHADED,            /* hits that reached triangle_interpolate */
    RT_CNT_TEX_FETCHES,       /* bilinear texture fetches */
    RT_CNT_PATHS,             /* camera samples finished */
    RT_CNT_MAX_STACK,         /* deepest traversal stack seen (only with RT_OPT_COUNTERS) */
    RT_CNT_WAVE_NODE_TRIPS,   /* inner-node loop trips executed per wave, summed (RT_OPT_COUNTERS): with BOX_TESTS/2 gives lane utilisation */
    RT_CNT_WAVE_LEAF_TRIPS,   /* leaf-phase trips executed per wave, summed (RT_OPT_COUNTERS) */
    RT_CNT_NODE_VISITS,       /* two-level nodes fetched, summed over lanes (RT_OPT_COUNTERS) */
    RT_CNT_HOT_VISITS,        /* of those, the ones served from the workgroup's LDS copy (RT_OPT_HOT_NODES) */
    RT_CNT_CYC_REFILL,        /* wavefront closest-hit, per wave and summed (RT_OPT_COUNTERS): clock cycles spent in refill trips, */
    RT_CNT_CYC_NODE,          /* ... in node trips, */
    RT_CNT_CYC_LEAF,          /* ... in leaf trips, */
    RT_CNT_CYC_WAVE,          /* ... and from a wave's first trip to its last (s_memtime, 100 MHz-independent shader clock) */
    RT_CNT_WAVE_SPAN_TICKS,   /* ... the same span in ticks of the constant 100 MHz clock, and */
    RT_CNT_WAVES,             /* the number of closest-hit waves that ran: with the kernel time, how full the chip was */
    RT_CNT_MAX_RAY_VISITS,    /* most node + leaf visits any single ray needed (RT_OPT_COUNTERS) */
    RT_CNT_NONFINITE_RAYS,    /* rays whose direction has a NaN or infinite component (RT_OPT_COUNTERS) */
    RT_CNT_LAUNCH_SLOT_TICKS, /* closest-hit launches (RT_OPT_COUNTERS): (last wave end - first wave start) x waves that worked, summed, 100 MHz ticks */
    RT_CNT_TAIL_IDLE_TICKS,   /* ... of which behind waves that ended before the launch's last one */
    RT_CNT_CYC_CLAIM,         /* closest-hit waves: clock cycles spent waiting for chunk claims (inside the refill cycles), and */
    RT_CNT_CLAIMS,            /* the number of claims */
    RT_CNT_WALKS,             /* closest-hit walks actually run: equals RT_CNT_RAYS unless samples share their first walk (RT_OPT_SHARE_PRIMARY) */
    RT_CNT_COUNT = 24
};

/* options for rt_set_option */
enum {
    RT_OPT_COUNTERS = 1,      /* 0 = rays/paths only (default), 1 = also box/triangle/shade/texture counts */
    RT_OPT_XCD_REMAP = 2,     /* 1 = give each XCD a contiguous band of pixel tiles (default), 0 = linear */
    RT_OPT_CHUNK = 3,         /* samples one lane traces in a row (default 8); scheduling only, results unchanged */
    RT_OPT_PASS = 4,          /* samples per pixel traced between two ordered sums (default 1024); scheduling only.  Megakernel: the
                                 context holds (pass + 1) * n_pixel * 12 bytes of sample slots (fewer samples when that would pass
                                 64 GiB).  Wavefront: see RT_OPT_SLOT_RING_MB */
    RT_OPT_LDS_SLOTS = 5,     /* traversal-stack levels kept in LDS (default 11); deeper levels spill to private memory */
    RT_OPT_MODE = 6,          /* 0 = megakernel: one kernel traces whole paths; 1 = wavefront (default): path state in HBM,
                                 one kernel per stage with queue compaction.  Same paths, same image, bit for bit.
                                 In wavefront mode rt_raytrace waits on the stream while it drives the stages. */
    RT_OPT_WF_CAPACITY_LOG2 = 7, /* wavefront pool size = at most 2^value paths (default 27; 7 * 16 bytes of HBM per path, 15 GB at the default), never more
                                 than a pass has paths; a long closest-hit launch hides the wait for its last long rays better than a short one */
    RT_OPT_WF_RUN = 8,        /* wavefront: pool entries a closest-hit wave claims from the queue at a time (default 256, >= 64); scheduling only */
    RT_OPT_WF_SORT = 10,      /* accepted and ignored (an earlier shading kernel filed its survivors by direction octant; the present one
                                 files them in lane order, which keeps the samples of a pixel next to each other) */
    RT_OPT_TIMING = 11,       /* 1 = bracket every traversal kernel launch with HIP events on the launch stream (see rt_get_timing) */
    RT_OPT_HOT_NODES = 12,    /* wavefront: node records nearest to the root that the closest-hit kernel copies to LDS (default 100, 0..512) */
    RT_OPT_WF_VOTE = 13,      /* wavefront closest-hit: 0 = fixed pattern (RT_OPT_WF_BURST node trips, then one leaf trip); k > 0 (default 32) = every trip
                                 the wave runs the visit kind most lanes wait for, a leaf visit weighing k/16 node visits; scheduling only */
    RT_OPT_WF_REFILL = 14,    /* with RT_OPT_WF_VOTE: idle lanes per wave that trigger a refill from the pool (default 24) */
    RT_OPT_WF_BLOCK = 15,     /* wavefront closest-hit: lanes per workgroup (default 768; a multiple of 64 up to 1024); scheduling only */
    RT_OPT_WF_LANES = 16,     /* wavefront: independent pipelines a pass's samples are split over, each on a stream of its own, so that
                                 one pipeline's kernels fill the CUs another's closest-hit launch has already left (default 1; 1 or 2; 2 measured slower);
                                 scheduling only */
    RT_OPT_WF_BURST = 9,      /* wavefront: inner-node trips between two looks at leaves and idle lanes (default 4); scheduling only */
    RT_OPT_HOT_TUNE = 18,     /* wavefront: 1 = before the first call of at least 64 samples per pixel after a scene upload or a camera
                                 change, one sample per pixel is traced to count the visits of the 4096 node records nearest the root, and
                                 those records are renumbered by falling count, so that the records the closest-hit kernel keeps in LDS
                                 (RT_OPT_HOT_NODES) are the ones this view enters most; 2 (default) = the visits are counted per band of
                                 the picture as well (a pass starts its paths tile row after tile row, so the paths in flight belong to
                                 one band at a time) and the closest-hit launches read those records from a copy numbered for the band
                                 in flight (up to 64 copies of 512 KB); 0 = static order (largest boxes first).
                                 Scheduling only: node numbers are addresses */
    RT_OPT_SHARE_PRIMARY = 19, /* wavefront: 1 (default) = the samples of a semantic batch, which upstream starts with one and the same jittered
                                 primary ray (cuda/raytrace.cu:210, its README's "Hints"), share ONE closest-hit walk for it: a new pool entry
                                 stands for a pixel's run of samples of one batch, is walked once, and every sample is shaded from that hit with
                                 its own draws.  Same image bit for bit, RT_CNT_RAYS unchanged (rays as upstream counts them), RT_CNT_WALKS and
                                 the box/triangle test counts drop.  0 = one walk per sample, as upstream executes it.  Off by itself when a run
                                 (the batch, or the call if that is shorter) is longer than an eighth of a pool region */
    RT_OPT_SHARE_EQUAL = 20,  /* wavefront: 1 (default) = samples of one pixel that arrive at a hit along one ray and leave a mirror, a pane of
                                 glass (the same way of its two) or a fuzz-free metal reflection leave along one ray again, bit for bit: it
                                 is walked once and every such sample is shaded from that hit.  Same image, RT_CNT_RAYS unchanged,
                                 RT_CNT_WALKS drops.  0 = every continued path is walked */
    RT_OPT_SLOT_RING_MB = 21, /* wavefront: MiB the sample slots may take (default 10240).  A pass starts its paths tile after tile (16x16
                                 pixels) and sums a tile's slots as soon as its last path has ended, so the slots are a ring of tiles, each
                                 pass * 256 * 12 bytes: as many as this allows, never more than the frame has, at least one.  A ring
                                 shorter than the tiles in flight (about 2400 at the default pool size) holds new paths back; results
                                 do not depend on it */
    RT_OPT_GROUP_SAMPLES = 22, /* wavefront with one walk per sample (RT_OPT_SHARE_PRIMARY off): 1 (default) = a tile's entries are ordered so that up to 16 samples
                                 of a pixel's semantic batch -- which start along one and the same primary ray -- are traced by
                                 neighbouring lanes (the vector L1 then serves four lanes that ask for one address in one clock instead
                                 of four); 0 = sample after sample, 256 pixels each.  Order of work only: same image, same counts */
    RT_OPT_GPU_BVH = 17       /* 0 (default): rt_upload_scene re-packs the caller's trees, visiting order and hit records are upstream's bit
                                 for bit.  1: it reads only the triangle arrays (objects[i].bvh may be null) and builds ONE tree over all
                                 objects on the device (agglomeration in Morton order, rocPRIM sort and scans; 30 M triangles in under 0.1 s),
                                 replacing upstream's host build (csrc/bvh.cpp:29-98); the tree is also the quicker one to walk.  Another tree means another visiting order: among equally
                                 near triangles another one may win, and the rare rays for which slab test and triangle test disagree may
                                 see another hit -- records then agree with an exhaustive search, not with upstream's walk */
};

const char* rt_last_error(void);
int rt_abi_version(void);

/* ---- host halves of the path ---- */

/* Replaces vrt::build_bvh (include/bvh.h:76, csrc/bvh.cpp:77-98): median split on
 * the longest centroid axis, <=2 triangles per leaf, post-order array, root last.
 * Returns the node count (>0), or RT_ERR_CAPACITY if capacity < count
 * (2*n_triangles always suffices). */
int rt_build_bvh(const rt_triangle* triangles, int n_triangles, rt_bvh_node* nodes_out, int capacity);

/* Replaces vrt::apply_transform (include/transform.cuh:16, cuda/transforms.cu:32-53):
 * position = R*(position*scale) + translation, normal = R*normal, in place. */
int rt_apply_transform(const rt_transform* transform, rt_triangle* triangles, int n_triangles);

/* ---- context ---- */
int rt_create(int device_id, rt_context** out);
int rt_destroy(rt_context* ctx);
int rt_set_option(rt_context* ctx, int option, int value);

/* Replaces the uploads in SceneLoader::load_scene (csrc/sceneloader.cpp:336-399):
 * takes the host arrays in upstream layout, re-packs them for 64-lane fetches
 * and copies them to the context's device.  May be called again to replace the scene. */
int rt_upload_scene(rt_context* ctx, const rt_object* objects, int n_objects,
                    const rt_material* materials, int n_materials,
                    const rt_texture* textures, int n_textures);

/* ---- the render calls Worker::render_ makes (csrc/worker.cpp:64-97) ---- */

/* Replaces vrt::init_randstate (include/camera.cuh:9, cuda/camera.cu:15-21).  The
 * generator is counter based, so there is no state array; this records the seed
 * (upstream uses time(0)) and the frame size. */
int rt_init_randstate(rt_context* ctx, uint64_t seed, int width, int height);

/* Replaces vrt::setup_raytrace (include/camera.cuh:11-16, cuda/camera.cu:43-64):
 * builds the camera frame on the host.  Primary rays are regenerated inside the
 * path kernel from (pixel, batch index), so no Ray array is written. */
int rt_setup_raytrace(rt_context* ctx, int width, int height, const float camera_pos[3],
                      const float camera_dir[3], const float camera_up[3], float fov_radians);

/* Replaces vrt::raytrace (include/raytrace.cuh:8-15, cuda/raytrace.cu:197-287) and the
 * batch loop around it (csrc/worker.cpp:71-91).  Traces global samples
 * [first_sample, first_sample + spp) of every pixel and adds them to
 * d_output (device, float[3*n_pixel], row-major, y up).  `semantic_batch` is
 * upstream's <batch_size>: samples k*B .. k*B+B-1 share one jittered primary
 * ray and are summed before being added to the buffer.  first_sample must be a
 * multiple of semantic_batch.  `stream` is a hipStream_t (may be null).  In megakernel mode the call only queues work on
 * `stream`.  In wavefront mode (the default) the host drives the stage kernels and WAITS inside the call, a few
 * iterations behind the device, until every pass but the tail of the last has drained; the result is complete once
 * `stream` is synchronised.  A context's working memory (sample slots, path pools, control block, counters) is reused by
 * every call: drive a context from one stream at a time, and synchronise that stream before using another. */
int rt_raytrace(rt_context* ctx, int n_pixel, int first_sample, int spp, int semantic_batch,
                int max_depth, const float ambient[3], float russian_roulette,
                float* d_output, void* stream);

/* rt_raytrace for a caller with one thread and something else to do.  Upstream's raytrace() is a kernel launch that returns
 * at once and Worker::render_ waits for it afterwards (csrc/worker.cpp:80-89); rt_raytrace in wavefront mode cannot return
 * before its passes have drained, because the host drives the stages.  rt_raytrace_start takes the same arguments (`ambient`
 * is copied), hands the call to a thread of the library and returns; rt_raytrace_wait joins that thread and returns the
 * call's status (rt_last_error then has its text in the waiting thread).  After a successful wait the result is complete once
 * `stream` is synchronised, as after rt_raytrace.  Between the two, the context and `stream` belong to the call: no other
 * entry point may be given this context (rt_raytrace_start itself answers RT_ERR_INVALID while a call is outstanding).
 * rt_raytrace_wait with nothing outstanding returns RT_OK; rt_destroy waits by itself. */
int rt_raytrace_start(rt_context* ctx, int n_pixel, int first_sample, int spp, int semantic_batch,
                      int max_depth, const float ambient[3], float russian_roulette,
                      float* d_output, void* stream);
int rt_raytrace_wait(rt_context* ctx);

/* No counterpart in the reference.  For callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  Same arguments, same result bit for bit as rt_raytrace,
 * but the call allocates, creates and frees nothing: the sample slots between the path kernel and the ordered sum
 * live in a static device array of RT_STATIC_SCRATCH_BYTES that every GPU the library runs on holds until the process
 * ends, every pass is one launch of the megakernel design whatever RT_OPT_MODE says, and no timing events are recorded.
 * The frame must fit twice: 2 * n_pixel * 12 bytes <= RT_STATIC_SCRATCH_BYTES (2 796 202 pixels), else RT_ERR_CAPACITY;
 * samples per pass = floor(RT_STATIC_SCRATCH_BYTES / (12 * n_pixel)) - 1 (one plane is the carry): 1 at 2000x1200, 57 at
 * 400x240.  All contexts of a device share that device's array and a call
 * leaves its slots behind: every slot and the carry plane are written before they are read, so nothing needs clearing,
 * but two calls that overlap in time on one GPU (two streams, two threads, two contexts) race.  Asynchronous on `stream`. */
#define RT_STATIC_SCRATCH_BYTES (64u << 20)
int rt_raytrace_static(rt_context* ctx, int n_pixel, int first_sample, int spp, int semantic_batch,
                       int max_depth, const float ambient[3], float russian_roulette,
                       float* d_output, void* stream);

/* Replaces vrt::post_process (include/raytrace.cuh:17, cuda/raytrace.cu:249-297):
 * c /= spp, clamp to [0,1], c = pow(c, 1/gamma), in place.  Asynchronous on `stream`
 * (upstream synchronises; callers that need that call rt_synchronize). */
int rt_post_process(rt_context* ctx, int n_pixels, int spp, float gamma, float* d_output, void* stream);

/* ---- parity / diagnostics entry points (no upstream counterpart: upstream has no tests) ---- */

/* The Ray array upstream's setup_raytrace_kernel would have written for semantic
 * batch `batch_index` (cuda/camera.cu:23-41): d_rays = float[6*width*height]. */
int rt_primary_rays(rt_context* ctx, uint32_t batch_index, float* d_rays, void* stream);

/* Closest hit of explicit rays (the object loop of cuda/raytrace.cu:214-219):
 * d_rays = float[6*n] (origin, direction); outputs per ray: object index, triangle
 * index within that object (both -1 on a miss), t (inf on a miss), u, v.  d_u/d_v may be null. */
int rt_closest_hit(rt_context* ctx, int n_rays, const float* d_rays, int32_t* d_object,
                   int32_t* d_triangle, float* d_t, float* d_u, float* d_v, void* stream);

int rt_get_counters(rt_context* ctx, uint64_t out[RT_CNT_COUNT]);   /* synchronises the device */
int rt_reset_counters(rt_context* ctx);                             /* also clears the timing totals */

/* With RT_OPT_TIMING: out[0],[1] = total milliseconds and launches of the wavefront closest-hit kernel (wf_extend),
 * out[2],[3] = of the wavefront shading kernel, out[4],[5] = of the megakernel (path_kernel), since the last
 * rt_reset_counters.  Measured with HIP events on the stream the kernels were launched on.  Synchronises the device. */
int rt_get_timing(rt_context* ctx, double out[8]);

/* Scene statistics after upload: out[0]=triangles, [1]=packed inner nodes, [2]=max stack depth,
 * [3]=device bytes of the packed scene, [4]=objects; with RT_OPT_GPU_BVH also [5]=microseconds the device build took,
 * [6]=microseconds the copy of the triangle arrays to the device took, [7]=levels of the binary hierarchy. */
int rt_scene_stats(rt_context* ctx, uint64_t out[8]);

/* Test aid, no upstream counterpart: reads the packed node records back and checks what the renumbering of the
 * LDS candidates (RT_OPT_HOT_TUNE) relies on.  out[0] = node records, out[1] = k = the records that may be renumbered
 * (the first min(4096, out[0])), out[2] = references held by a record >= k to a record < k (must be 0: such a
 * reference would not follow a renumbering), out[3] = references to a record number >= out[0] (must be 0),
 * out[4] = inner references to a record whose number is not above the referring record's (must be 0: parents carry
 * lower numbers than their children).  Copies the whole node array to the host: for tests on small scenes. */
int rt_check_node_numbering(rt_context* ctx, uint64_t out[8]);

/* ---- multi-GPU: sample split and the sum onto the root GPU ----
 * No upstream counterpart (upstream is single-GPU, no collective anywhere).  The frame splits where upstream's batch
 * loop runs (csrc/worker.cpp:71-91): batches are independent, so each GPU traces a contiguous run of whole batches of
 * every pixel (scene replicated), and the float buffers are added onto the root before rt_post_process with the total
 * spp.  The random streams are keyed by (pixel, global sample), so the set of samples is the same for any GPU count;
 * only the order of the float additions across GPUs differs. */

int rt_get_device(rt_context* ctx, int* device);

/* Part `part` of `parts`: global samples [*first_sample, *first_sample + *count), whole batches except where the frame ends. */
int rt_partition_samples(int n_samples, int semantic_batch, int parts, int part, int* first_sample, int* count);

typedef struct rt_group rt_group;
#define RT_GROUP_ID_BYTES 128
/* ncclGetUniqueId: call in one process, hand the bytes to the others (any channel), pass to rt_group_create. */
int rt_group_unique_id(uint8_t id[RT_GROUP_ID_BYTES]);
/* The contexts this process drives (n_local >= 1; one host thread per GPU, or one process per GPU with n_local = 1).
 * Their distinct devices take RCCL ranks first_rank, first_rank + 1, ... of a communicator of `world` ranks; contexts
 * that share a device share its rank.  id = null: the world is this process alone (ncclCommInitAll), first_rank must
 * then be 0. */
int rt_group_create(rt_context* const* ctxs, int n_local, int first_rank, int world, const uint8_t* id, rt_group** out);
int rt_group_destroy(rt_group* group);
/* Adds up d_buffers[k] (device, `count` floats, k-th context of rt_group_create, its work queued on streams[k]; streams
 * may be null = null streams) over the whole world: contexts that share a device are added on the device in the given
 * order, then ONE ncclReduce (float, sum, root 0) runs over the communicator.  The sum lands in the buffer of the
 * context that leads RCCL rank 0 (d_buffers[0] of the process with first_rank 0); other buffers are left modified.
 * Asynchronous on the leaders' streams. */
int rt_reduce_to_root(rt_group* group, float* const* d_buffers, uint64_t count, void* const* streams);

/* ---- device-memory helpers for callers that do not link the HIP runtime ---- */
int rt_device_alloc(rt_context* ctx, uint64_t bytes, void** d_ptr);
int rt_device_free(rt_context* ctx, void* d_ptr);
int rt_device_zero(rt_context* ctx, void* d_ptr, uint64_t bytes, void* stream);
int rt_copy_to_host(rt_context* ctx, void* h_dst, const void* d_src, uint64_t bytes);
int rt_copy_to_device(rt_context* ctx, void* d_dst, const void* h_src, uint64_t bytes);
int rt_synchronize(rt_context* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RTCORE_H */
