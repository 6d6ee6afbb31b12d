// How much straight-line code a loop may hold before it streams: loops of KB kilobytes of 8-byte VALU instructions, two waves per SIMD
// (the occupancy of the pairing kernels), every CU running the same code.  hipcc --offload-arch=gfx950 -O2 -o ubench_icache ubench_icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;
#define STR2(x) #x
#define STR(x) STR2(x)
template <int KB> __global__ void __launch_bounds__(64) k_body(u32* out, int iters, u32 seed) {
    u32 r0 = seed + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#define BODY(N) asm volatile(".rept " STR(N) "\n v_add_u32 %0, 0x12345, %0\n .endr" : "+v"(r0));
        if (KB == 16) BODY(2048) if (KB == 32) BODY(4096) if (KB == 40) BODY(5120) if (KB == 48) BODY(6144) if (KB == 56) BODY(7168)
        if (KB == 60) BODY(7680) if (KB == 64) BODY(8192) if (KB == 68) BODY(8704) if (KB == 72) BODY(9216) if (KB == 80) BODY(10240)
        if (KB == 96) BODY(12288)
    }
    out[blockIdx.x * 64 + threadIdx.x] = r0;
}
template <int KB> void run(u32* out, int ncu) {
    const int iters = (1 << 22) / (KB * 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_body<KB>, dim3(ncu * 8), dim3(64), 0, 0, out, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double instr_per_simd = 2.0 * iters * KB * 128;                 // two waves per SIMD
    printf("ICACHE body=%3d KB  %8.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", KB, best, best * 1e-3 * 2.4e9 / instr_per_simd);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    u32* out; hipMalloc(&out, sizeof(u32) * p.multiProcessorCount * 8 * 64);
    const int n = p.multiProcessorCount;
    run<16>(out, n); run<32>(out, n); run<40>(out, n); run<48>(out, n); run<56>(out, n); run<60>(out, n); run<64>(out, n); run<68>(out, n);
    run<72>(out, n); run<80>(out, n); run<96>(out, n);
    return 0;
}
