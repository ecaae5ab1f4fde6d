// rt_multi.hip -- the multi-GPU half of the C ABI: sample partition, and the sum of the per-GPU accumulation buffers
// onto the root GPU with one RCCL reduce over xGMI.
//
// Upstream is single-GPU; the place where its frame splits is the batch loop of Worker::render_
// (csrc/worker.cpp:71-91): batches are independent, so each GPU traces a contiguous run of whole batches of every
// pixel with the scene replicated, and the float buffers are added together before post_process.
//
// RCCL is bound at run time (dlopen of librccl.so.1), so the library loads on machines without it and a process that
// already holds a copy -- PyTorch ships one under the same soname -- keeps using that copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtcore.h"

extern "C" int rt_set_last_error_(int code, const char* message);   // rt_kernels.hip: records the message for rt_last_error()

namespace {

int fail(int code, const std::string& msg) { return rt_set_last_error_(code, msg.c_str()); }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl() {
    static Rccl* lib = [] {
        Rccl* r = new Rccl();
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r->handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r->handle) break;
        }
        if (!r->handle) { r->error = std::string("RCCL is not available: ") + dlerror(); return r; }
        bool ok = true;
        auto sym = [&](const char* n) { void* p = dlsym(r->handle, n); if (!p) { ok = false; r->error = std::string("RCCL lacks ") + n; } return p; };
        r->GetUniqueId = (decltype(r->GetUniqueId))sym("ncclGetUniqueId");
        r->CommInitRank = (decltype(r->CommInitRank))sym("ncclCommInitRank");
        r->CommInitAll = (decltype(r->CommInitAll))sym("ncclCommInitAll");
        r->CommDestroy = (decltype(r->CommDestroy))sym("ncclCommDestroy");
        r->Reduce = (decltype(r->Reduce))sym("ncclReduce");
        r->GroupStart = (decltype(r->GroupStart))sym("ncclGroupStart");
        r->GroupEnd = (decltype(r->GroupEnd))sym("ncclGroupEnd");
        r->GetErrorString = (decltype(r->GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(r->handle); r->handle = nullptr; }
        return r;
    }();
    return lib;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess)                                                                           \
            return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + rccl()->GetErrorString(r_));          \
    } while (0)

// dst += src, element by element (one addition per element: the order of a pixel's terms is the order of the calls)
__global__ __launch_bounds__(256) void add_into_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = dst[i] + src[i];
}

}  // namespace

// One RCCL rank per distinct device of this process (RCCL allows a device once per communicator); contexts that share
// a device are summed on that device before the collective.
struct rt_group {
    struct Member { int device; int slot; hipEvent_t ready; };   // slot = index into `devices`
    std::vector<Member> members;                                  // one per context, in the caller's order
    std::vector<int> devices;                                     // distinct devices, in order of first use
    std::vector<ncclComm_t> comms;                                // one per distinct device
    int first_rank = 0, world = 1;
};

extern "C" {

int rt_partition_samples(int n_samples, int semantic_batch, int parts, int part, int* first_sample, int* count) {
    if (n_samples < 0 || semantic_batch <= 0 || parts <= 0 || part < 0 || part >= parts || !first_sample || !count)
        return fail(RT_ERR_INVALID, "rt_partition_samples: bad argument");
    // contiguous runs of whole batches, the remainder batches to the low parts; only the run that ends the frame may end mid-batch
    const long long n_batches = ((long long)n_samples + semantic_batch - 1) / semantic_batch;
    const long long base = n_batches / parts, extra = n_batches % parts;
    const long long first_batch = part * base + (part < extra ? part : extra);
    const long long mine = base + (part < extra ? 1 : 0);
    const long long first = first_batch * semantic_batch;
    long long end = (first_batch + mine) * semantic_batch;
    if (end > n_samples) end = n_samples;
    *first_sample = (int)first;
    *count = end > first ? (int)(end - first) : 0;
    return RT_OK;
}

int rt_group_unique_id(uint8_t id[RT_GROUP_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) == RT_GROUP_ID_BYTES, "RT_GROUP_ID_BYTES");
    if (!id) return fail(RT_ERR_INVALID, "rt_group_unique_id: null argument");
    if (!rccl()->handle) return fail(RT_ERR_DEVICE, rccl()->error);
    ncclUniqueId u;
    NCCL_TRY(rccl()->GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return RT_OK;
}

int rt_group_create(rt_context* const* ctxs, int n_local, int first_rank, int world, const uint8_t* id, rt_group** out) {
    if (!ctxs || n_local <= 0 || !out || first_rank < 0 || world <= 0) return fail(RT_ERR_INVALID, "rt_group_create: bad argument");
    if (!rccl()->handle) return fail(RT_ERR_DEVICE, rccl()->error);
    rt_group* g = new rt_group();
    g->first_rank = first_rank;
    g->world = world;
    for (int k = 0; k < n_local; k++) {
        int dev = -1;
        if (!ctxs[k] || rt_get_device(ctxs[k], &dev) != RT_OK) { delete g; return fail(RT_ERR_INVALID, "rt_group_create: null context"); }
        int slot = -1;
        for (size_t d = 0; d < g->devices.size(); d++) if (g->devices[d] == dev) slot = (int)d;
        if (slot < 0) { slot = (int)g->devices.size(); g->devices.push_back(dev); }
        g->members.push_back(rt_group::Member{dev, slot, nullptr});
    }
    const int n_dev = (int)g->devices.size();
    if (first_rank + n_dev > world) { delete g; return fail(RT_ERR_INVALID, "rt_group_create: more local devices than ranks left in the world"); }
    // without an id the communicator is made by ncclCommInitAll, which numbers this process's devices from rank 0
    if (!id && first_rank != 0) { delete g; return fail(RT_ERR_INVALID, "rt_group_create: first_rank must be 0 when the world lives in one process (no id)"); }
    if (!id && n_dev != world) { delete g; return fail(RT_ERR_INVALID, "rt_group_create: a world that spans processes needs the id of rt_group_unique_id"); }
    for (rt_group::Member& m : g->members) {
        hipError_t e = hipSetDevice(m.device);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m.ready, hipEventDisableTiming);
        if (e != hipSuccess) { rt_group_destroy(g); return fail(RT_ERR_DEVICE, std::string("rt_group_create: ") + hipGetErrorString(e)); }
    }
    g->comms.assign(n_dev, nullptr);
    ncclResult_t r = ncclSuccess;
    if (!id) {
        r = rccl()->CommInitAll(g->comms.data(), n_dev, g->devices.data());
    } else {
        ncclUniqueId u;
        std::memcpy(&u, id, sizeof u);
        r = rccl()->GroupStart();
        for (int d = 0; d < n_dev && r == ncclSuccess; d++) {
            if (hipSetDevice(g->devices[d]) != hipSuccess) { r = ncclUnhandledCudaError; break; }
            r = rccl()->CommInitRank(&g->comms[d], world, u, first_rank + d);
        }
        const ncclResult_t r2 = rccl()->GroupEnd();
        if (r == ncclSuccess) r = r2;
    }
    if (r != ncclSuccess) {
        const std::string msg = std::string("rt_group_create: RCCL communicator: ") + rccl()->GetErrorString(r);
        rt_group_destroy(g);
        return fail(RT_ERR_DEVICE, msg);
    }
    *out = g;
    return RT_OK;
}

int rt_group_destroy(rt_group* g) {
    if (!g) return RT_OK;
    for (size_t d = 0; d < g->comms.size(); d++)
        if (g->comms[d]) { (void)hipSetDevice(g->devices[d]); (void)rccl()->CommDestroy(g->comms[d]); }
    for (rt_group::Member& m : g->members)
        if (m.ready) { (void)hipSetDevice(m.device); (void)hipEventDestroy(m.ready); }
    delete g;
    return RT_OK;
}

int rt_reduce_to_root(rt_group* g, float* const* d_buffers, uint64_t count, void* const* streams) {
    if (!g || !d_buffers) return fail(RT_ERR_INVALID, "rt_reduce_to_root: null argument");
    const size_t n = g->members.size();
    for (size_t k = 0; k < n; k++) if (!d_buffers[k]) return fail(RT_ERR_INVALID, "rt_reduce_to_root: null buffer");
    if (count == 0) return RT_OK;
    // per device: the first context's buffer and stream lead; the others' buffers are added to it in the caller's order
    std::vector<int> leader(g->devices.size(), -1);
    for (size_t k = 0; k < n; k++) {
        const rt_group::Member& m = g->members[k];
        hipStream_t s = streams ? (hipStream_t)streams[k] : nullptr;
        if (leader[m.slot] < 0) { leader[m.slot] = (int)k; continue; }
        const int lead = leader[m.slot];
        hipStream_t ls = streams ? (hipStream_t)streams[lead] : nullptr;
        HIP_TRY(hipSetDevice(m.device));
        HIP_TRY(hipEventRecord(m.ready, s));
        HIP_TRY(hipStreamWaitEvent(ls, m.ready, 0));
        const unsigned blocks = (unsigned)std::min<uint64_t>((count + 255) / 256, 256u * 64u);
        hipLaunchKernelGGL(add_into_kernel, dim3(blocks), dim3(256), 0, ls, d_buffers[lead], (const float*)d_buffers[k], (size_t)count);
        HIP_TRY(hipGetLastError());
    }
    // one reduce over the communicator: every rank contributes its leader buffer, rank 0 receives the sum in place
    NCCL_TRY(rccl()->GroupStart());
    ncclResult_t r = ncclSuccess;
    for (size_t d = 0; d < g->devices.size() && r == ncclSuccess; d++) {
        const int lead = leader[d];
        if (hipSetDevice(g->devices[d]) != hipSuccess) { r = ncclUnhandledCudaError; break; }
        r = rccl()->Reduce(d_buffers[lead], d_buffers[lead], (size_t)count, ncclFloat32, ncclSum, 0, g->comms[d],
                           streams ? (hipStream_t)streams[lead] : nullptr);
    }
    const ncclResult_t r2 = rccl()->GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess) return fail(RT_ERR_DEVICE, std::string("rt_reduce_to_root: ncclReduce: ") + rccl()->GetErrorString(r));
    return RT_OK;
}

}  // extern "C"
