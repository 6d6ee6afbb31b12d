// ubench_sort.hip -- is a device radix sort (rocPRIM through hipCUB) a faster way to group the MSM's 2^24 (window, digit) -> point items than
// the one-pass atomic scatter of msm.inc (k_msm_scatter_cap: 0.99 ms per 2^20 points)?   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_sort tools/ubench_sort.hip
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main() {
    const size_t n = (size_t)16 << 20;
    std::vector<unsigned> k(n), v(n);
    std::mt19937 rng(1);
    for (size_t i = 0; i < n; i++) { k[i] = rng() & 0x7ffff; v[i] = (unsigned)(i >> 4); }
    unsigned *dk, *dv, *dk2, *dv2;
    hipMalloc(&dk, 4 * n); hipMalloc(&dv, 4 * n); hipMalloc(&dk2, 4 * n); hipMalloc(&dv2, 4 * n);
    hipMemcpy(dk, k.data(), 4 * n, hipMemcpyHostToDevice); hipMemcpy(dv, v.data(), 4 * n, hipMemcpyHostToDevice);
    for (int bits : {19, 16, 32}) {
        size_t tb = 0;
        hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dk2, dv, dv2, (int)n, 0, bits);
        void* tmp; hipMalloc(&tmp, tb);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float best = 1e9;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(a);
            hipcub::DeviceRadixSort::SortPairs(tmp, tb, dk, dk2, dv, dv2, (int)n, 0, bits);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("SortPairs 2^24 u32/u32, %d key bits: %.3f ms (temp %.1f MB)\n", bits, best, tb / 1048576.0);
        hipFree(tmp);
    }
    return 0;
}
