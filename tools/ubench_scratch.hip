// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on a KNOWN per-lane scratch pattern ("calibrate FETCH_SIZE on
// a known scratch pattern"): every lane owns a 4 KB private array, writes all of it and reads all of it back `rounds` times through
// run-time indices (so it lives in scratch memory, not registers).  8 192 waves x 64 lanes x 4 KB = 2 GB of footprint launched,
// 512 MB resident at a time -- beyond L2 (32 MB) and the Infinity Cache (256 MB).  Algorithmic bytes: lanes x rounds x 4 096 each way.
// hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench_scratch tools/ubench_scratch.hip; run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64, 2) k_scratch_pattern(int* out, int rounds, int salt) {
    int a[1024];
    const int tid = threadIdx.x + 64 * blockIdx.x;
    int sum = 0;
    for (int r = 0; r < rounds; r++) {
        for (int i = 0; i < 1024; i++) a[(i * 17 + salt) & 1023] = i + r + tid;
        for (int i = 0; i < 1024; i++) sum += a[(i * 29 + salt + r) & 1023];
    }
    out[tid] = sum;
}
int main(int argc, char** argv) {
    const int waves = 8192, rounds = argc > 1 ? atoi(argv[1]) : 4;
    int* d; hipMalloc(&d, waves * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_scratch_pattern, dim3(waves), dim3(64), 0, 0, d, 1, 3); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k_scratch_pattern, dim3(waves), dim3(64), 0, 0, d, rounds, 5); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)waves * 64 * rounds * 4096;
    printf("k_scratch_pattern: %d waves, %d rounds: %.3f ms; algorithmic scratch bytes written %.3f GB, read %.3f GB -> %.0f GB/s each way\n", waves, rounds, ms, bytes / 1e9, bytes / 1e9, bytes / 1e6 / ms);
    return 0;
}
