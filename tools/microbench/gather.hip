// Micro-benchmark: how fast can a CU gather 64-byte records at per-lane random indices, and does it matter whether
// each lane fetches its own record (4 x 16 B) or the four lanes of a quad fetch each record together (16 B each)?
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/microbench/gather tools/microbench/gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

// mode 0: lane-private 64 B (4 x dwordx4).  mode 1: quad-cooperative (4 instr, each quad loads one lane's record, DPP transpose skipped:
// we only sum what we loaded, the point is the memory pipeline).  mode 2: lane-private 16 B only.  mode 3: lane-private 128 B.
template <int MODE>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ tab, uint32_t mask, int iters, float* out, int dependent) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    uint32_t idx = lcg(s) & mask;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            const float4* p = tab + (size_t)idx * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
        } else if (MODE == 1) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t nj;
                if (j == 0) nj = __builtin_amdgcn_mov_dpp((int)idx, 0x00, 0xf, 0xf, true);
                if (j == 1) nj = __builtin_amdgcn_mov_dpp((int)idx, 0x55, 0xf, 0xf, true);
                if (j == 2) nj = __builtin_amdgcn_mov_dpp((int)idx, 0xaa, 0xf, 0xf, true);
                if (j == 3) nj = __builtin_amdgcn_mov_dpp((int)idx, 0xff, 0xf, 0xf, true);
                v[j] = tab[(size_t)nj * 4 + (lane & 3)];
            }
            acc += v[0].x + v[1].y + v[2].z + v[3].w;
        } else if (MODE == 2) {
            const float4 a = tab[(size_t)idx * 4];
            acc += a.x;
        } else {
            const float4* p = tab + (size_t)(idx & ~1u) * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6], h = p[7];
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
        }
        // next index: independent of the data, or (dependent) a function of it, as in a tree walk
        uint32_t nx = lcg(s);
        if (dependent) nx ^= (uint32_t)(acc != 12345.678f ? 0 : 1);
        idx = nx & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, const float4* tab, uint32_t mask, float* out, int blocks, int iters, int dep) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, tab, mask, 8, out, dep);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out, dep);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double recs = (double)blocks * 256 * iters;
    printf("%-28s table %6.1f MB dep %d: %7.3f ms  %7.2f G records/s  %6.3f records/clk/CU @2.4GHz\n", name,
           (mask + 1.0) * 64 / 1e6, dep, ms, recs / ms / 1e6, recs / (ms * 1e-3) / 256 / 2.4e9);
}

int main(int argc, char** argv) {
    const int blocks = 256 * 8;
    // optional arguments: largest table as log2 of its record count (default 24 = 1 GB), iterations per lane (default 2000)
    const int max_log2n = argc > 1 ? atoi(argv[1]) : 24;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    for (int log2n = 14; log2n <= max_log2n; log2n += (log2n < 20 ? 3 : 2)) {
        const uint32_t n = 1u << log2n;
        float4* tab; float* out;
        CHECK(hipMalloc(&tab, (size_t)n * 64 + 256));
        CHECK(hipMemset(tab, 0, (size_t)n * 64 + 256));
        CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
        for (int dep = 0; dep < 2; dep++) {
            run<0>("lane-private 4x16B", tab, n - 1, out, blocks, iters, dep);
            run<1>("quad-cooperative 4x16B", tab, n - 1, out, blocks, iters, dep);
            run<2>("lane-private 1x16B", tab, n - 1, out, blocks, iters, dep);
            run<3>("lane-private 8x16B (128B)", tab, n - 1, out, blocks, iters, dep);
        }
        CHECK(hipFree(tab)); CHECK(hipFree(out));
    }
    return 0;
}
