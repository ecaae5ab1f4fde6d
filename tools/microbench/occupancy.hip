// occupancy.hip -- how many waves of a kernel does a CU of this GPU really hold?  Every wave notes when it starts and
// ends (100 MHz wall clock) and spins for a while; the host sweeps the stamps for the largest number of waves that
// overlapped.  Variants: workgroup size, dynamic LDS per workgroup, with or without a private (scratch) array.
//   hipcc -O3 --offload-arch=gfx950 -o occupancy occupancy.hip && ./occupancy
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

template <int SCRATCH, int REGS>
__global__ __launch_bounds__(1024) void spin(unsigned long long* stamps, int spin_ticks, unsigned* sink) {
    extern __shared__ unsigned lds[];
    unsigned priv[SCRATCH ? SCRATCH : 1];
    const unsigned long long t0 = wall_clock64();
    unsigned acc = threadIdx.x;
    if (REGS == 1) asm volatile("" ::: "v60", "s89");            // the register budget of the closest-hit kernel: 61 VGPRs, 90 SGPRs
    if (REGS == 2) asm volatile("" ::: "v60");
    if (REGS == 3) asm volatile("" ::: "s89");
    if (SCRATCH) {
        for (int i = 0; i < SCRATCH; i++) priv[i] = acc * (i + 1);
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) {
        acc = acc * 1664525u + 1013904223u;
        if (SCRATCH) { priv[acc % (SCRATCH ? SCRATCH : 1)] += acc; acc += priv[(acc >> 8) % (SCRATCH ? SCRATCH : 1)]; }
    }
    const unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t0;
        stamps[2 * w + 1] = t1;
    }
    if (acc == 0xFFFFFFFFu) *sink = acc + lds[(threadIdx.x + 1) % blockDim.x];
}

template <int SCRATCH, int REGS>
static void run(int block, int lds_bytes, int groups_per_cu, int n_cu) {
    const int grid = n_cu * groups_per_cu, waves = grid * (block / 64);
    unsigned long long* d;
    unsigned* sink;
    hipMalloc(&d, sizeof(unsigned long long) * 2 * waves);
    hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void*)spin<SCRATCH, REGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL((spin<SCRATCH, REGS>), dim3(grid), dim3(block), lds_bytes, 0, d, 20000 /* 200 us */, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
    std::vector<std::pair<unsigned long long, int>> ev;
    for (int w = 0; w < waves; w++) { ev.push_back({h[2 * w], 1}); ev.push_back({h[2 * w + 1], -1}); }
    std::sort(ev.begin(), ev.end());
    int cur = 0, best = 0;
    for (auto& e : ev) { cur += e.second; best = std::max(best, cur); }
    printf("block %4d  LDS %6d B/group  %d groups/CU offered (%5d waves)  scratch %d regs %d : at most %5d waves at once = %.1f per CU = %.2f per SIMD\n", block,
           lds_bytes, groups_per_cu, waves, (int)SCRATCH, REGS, best, (double)best / n_cu, (double)best / n_cu / 4);
    hipFree(d);
    hipFree(sink);
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%d CUs\n", n_cu);
    auto sizes = [&](auto go) {
        go(1024, 77776, 2);
        go(768, 79824, 2);
        go(896, 76752, 2);
        go(512, 40912, 4);
        go(256, 4096, 8);
    };
    // regs: 0 = as few as it takes, 1 = 61 VGPRs and 90 SGPRs (the closest-hit kernel's), 2 = the VGPRs only, 3 = the SGPRs only
    sizes([&](int b, int l, int g) { run<0, 0>(b, l, g, n_cu); });
    sizes([&](int b, int l, int g) { run<116, 0>(b, l, g, n_cu); });
    sizes([&](int b, int l, int g) { run<0, 1>(b, l, g, n_cu); });
    sizes([&](int b, int l, int g) { run<0, 2>(b, l, g, n_cu); });
    sizes([&](int b, int l, int g) { run<0, 3>(b, l, g, n_cu); });
    sizes([&](int b, int l, int g) { run<116, 1>(b, l, g, n_cu); });
    return 0;
}
