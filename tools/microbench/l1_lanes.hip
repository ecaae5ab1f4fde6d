// Micro-benchmark: what does a wave-level gather of 128-byte records (seven 16-byte pieces per lane, as a node visit of the
// closest-hit kernel reads them) cost the vector L1 as a function of HOW MANY LANES take part?  If the cost follows the active
// lanes, filling idle lanes buys the L1 nothing; if it is per instruction, it buys everything.
// Tables: 16 KB (lives in the 32 KB L1), 1 MB (L2), 64 MB (Infinity Cache / HBM).  Lanes: the first K of every wave, or K lanes spread out.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/microbench/l1_lanes tools/microbench/l1_lanes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

template <int PIECES>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ tab, uint32_t mask, int iters, float* out, unsigned long long lanes) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    const bool active = (lanes >> lane) & 1ull;
    uint32_t idx = lcg(s) & mask;
    for (int it = 0; it < iters; it++) {
        if (active) {
            const float4* p = tab + (size_t)idx * 8;
            float4 v[PIECES];
#pragma unroll
            for (int k = 0; k < PIECES; k++) v[k] = p[k];
#pragma unroll
            for (int k = 0; k < PIECES; k++) acc += v[k].x;
        }
        idx = (lcg(s) ^ (uint32_t)(acc == 12345.678f)) & mask;     // the next record depends on the data, as in a tree walk
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// Cooperative variants: the lanes of a group of G (2, 4 or 8) read ONE record together, piece (lane % G) [+ G, ...] each, one
// group member's record per instruction: the same number of lane accesses per record as the lane-private gather when G
// divides 8, but the lanes of a group touch one 128-byte line per instruction.  Does the L1 charge per lane or per line?
template <int G>
__global__ __launch_bounds__(256) void gather_coop(const float4* __restrict__ tab, uint32_t mask, int iters, float* out) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    uint32_t idx = lcg(s) & mask;
    for (int it = 0; it < iters; it++) {
        float4 v[8];
#pragma unroll
        for (int m = 0; m < G; m++) {                       // member m's record, read by the whole group
            const uint32_t rec = (uint32_t)__shfl((int)idx, (lane & ~(G - 1)) + m, 64);
#pragma unroll
            for (int k = 0; k < 8 / G; k++) v[m * (8 / G) + k] = tab[(size_t)rec * 8 + (lane & (G - 1)) + k * G];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k].x;
        idx = (lcg(s) ^ (uint32_t)(acc == 12345.678f)) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// Duplicate addresses: the lanes of a group of G read the SAME pieces of the SAME record (rays that coincide walk the same
// nodes in lockstep).  Does the L1 charge such a load per lane or per distinct address?
template <int G>
__global__ __launch_bounds__(256) void gather_dup(const float4* __restrict__ tab, uint32_t mask, int iters, float* out) {
    uint32_t s = (blockIdx.x * 256 + (threadIdx.x & ~(G - 1))) * 2654435761u + 12345u;       // one random stream per group
    float acc = 0.0f;
    uint32_t idx = lcg(s) & mask;
    for (int it = 0; it < iters; it++) {
        const float4* p = tab + (size_t)idx * 8;
        float4 v[7];
#pragma unroll
        for (int k = 0; k < 7; k++) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < 7; k++) acc += v[k].x;
        idx = (lcg(s) ^ (uint32_t)(acc == 12345.678f)) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int G>
static void run_dup(const float4* tab, uint32_t mask, float* out, int blocks, int iters) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather_dup<G>, dim3(blocks), dim3(256), 0, 0, tab, mask, 8, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(gather_dup<G>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr = (double)blocks * 4 * iters * 7;
    const double clk = ms * 1e-3 * 2.4e9 * 256;
    printf("table %8.3f MB 7 pieces, groups of %2d lanes read the same addresses: %8.3f ms  %6.2f clk per wave instruction (%d distinct addresses)\n",
           (mask + 1.0) * 128 / 1e6, G, ms, clk / wave_instr, 64 / G);
}

template <int G>
static void run_coop(const float4* tab, uint32_t mask, float* out, int blocks, int iters) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather_coop<G>, dim3(blocks), dim3(256), 0, 0, tab, mask, 8, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(gather_coop<G>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr = (double)blocks * 4 * iters * 8;
    const double clk = ms * 1e-3 * 2.4e9 * 256;
    printf("table %8.3f MB 8 pieces, groups of %d lanes share a record: %8.3f ms  %6.3f lane accesses/clk/CU  %6.2f clk per wave instruction  %6.3f records/clk/CU\n",
           (mask + 1.0) * 128 / 1e6, G, ms, wave_instr * 64 / clk, clk / wave_instr, (double)blocks * 256 * iters / clk);
}

template <int PIECES>
static void run(const float4* tab, uint32_t mask, float* out, int blocks, int iters, unsigned long long lanes, const char* what) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather<PIECES>, dim3(blocks), dim3(256), 0, 0, tab, mask, 8, out, lanes);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(gather<PIECES>, dim3(blocks), dim3(256), 0, 0, tab, mask, iters, out, lanes);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const int k = __builtin_popcountll(lanes);
    const double wave_instr = (double)blocks * 4 * iters * PIECES;          // wave-level load instructions
    const double lane_acc = wave_instr * k;
    const double clk = ms * 1e-3 * 2.4e9 * 256;                             // CU-clocks
    printf("table %8.3f MB pieces %d lanes %2d %-9s: %8.3f ms  %6.3f lane accesses/clk/CU  %6.2f clk per wave instruction\n",
           (mask + 1.0) * 128 / 1e6, PIECES, k, what, ms, lane_acc / clk, clk / wave_instr);
}

int main(int argc, char** argv) {
    const int blocks = 256 * 8;
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    for (int log2n : {7, 13, 19}) {
        const uint32_t n = 1u << log2n;
        float4* tab; float* out;
        CHECK(hipMalloc(&tab, (size_t)n * 128 + 256));
        CHECK(hipMemset(tab, 0, (size_t)n * 128 + 256));
        CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
        for (int k : {16, 32, 48, 64}) {
            const unsigned long long first = k == 64 ? ~0ull : ((1ull << k) - 1ull);
            unsigned long long spread = 0;
            for (int i = 0; i < k; i++) spread |= 1ull << (i * 64 / k);
            run<7>(tab, n - 1, out, blocks, iters, first, "first");
            if (k < 64) run<7>(tab, n - 1, out, blocks, iters, spread, "spread");
        }
        run_dup<1>(tab, n - 1, out, blocks, iters);
        run_dup<2>(tab, n - 1, out, blocks, iters);
        run_dup<4>(tab, n - 1, out, blocks, iters);
        run_dup<16>(tab, n - 1, out, blocks, iters);
        run_dup<64>(tab, n - 1, out, blocks, iters);
        run_coop<1>(tab, n - 1, out, blocks, iters);
        run_coop<2>(tab, n - 1, out, blocks, iters);
        run_coop<4>(tab, n - 1, out, blocks, iters);
        run_coop<8>(tab, n - 1, out, blocks, iters);
        run<1>(tab, n - 1, out, blocks, iters, ~0ull, "first");
        run<3>(tab, n - 1, out, blocks, iters, ~0ull, "first");
        CHECK(hipFree(tab)); CHECK(hipFree(out));
    }
    return 0;
}
