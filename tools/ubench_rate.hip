// Issue rate of the VALU instructions the field arithmetic is made of, at the occupancy of the pairing kernels (two waves per SIMD,
// 64-thread workgroups, eight per CU).  Each wave runs a chain on ONE register (dependent) or on FOUR registers round-robin (independent).
// hipcc --offload-arch=gfx950 -O2 -w -o ubench_rate ubench_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef unsigned long long u64;
#define STR2(x) #x
#define STR(x) STR2(x)
#define N 1024
template <int OP, int WIDE> __global__ void __launch_bounds__(64) k_rate(u32* out, int iters, u32 seed) {
    u32 a = seed + threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 77, m = threadIdx.x | 1;
    u64 A = a, B = b, C = c, D = d;
    for (int it = 0; it < iters; ++it) {
#define CH1(ins) asm volatile(".rept " STR(N) "\n " ins "\n .endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(m) : "vcc");
        if (!WIDE) {
            if (OP == 0) CH1("v_add_u32 %0, 0x12345, %0")
            if (OP == 1) CH1("v_add_u32 %0, %8, %0")
            if (OP == 2) CH1("v_and_b32 %0, %8, %0")
            if (OP == 3) CH1("v_lshrrev_b32 %0, 1, %0")
            if (OP == 4) CH1("v_add3_u32 %0, %0, %8, %8")
            if (OP == 5) CH1("v_mul_lo_u32 %0, %0, %8")
            if (OP == 6) CH1("v_mad_u64_u32 %4, vcc, %0, %8, %4")
            if (OP == 7) CH1("v_mad_i64_i32 %4, vcc, %0, %8, %4")
            if (OP == 8) CH1("v_ashrrev_i64 %4, 1, %4")
            if (OP == 9) CH1("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
            if (OP == 10) CH1("v_bfi_b32 %0, %8, %0, %0")
            if (OP == 11) CH1("v_sub_u32 %0, %0, %8")
            if (OP == 12) CH1("v_mov_b32 %0, %1\n v_mov_b32 %1, %0")
            if (OP == 13) CH1("v_and_or_b32 %0, %0, %8, %8")
            if (OP == 14) CH1("v_alignbit_b32 %0, %0, %8, 27")
            if (OP == 15) CH1("v_lshl_add_u32 %0, %0, 1, %8")
        } else {
            if (OP == 0) CH1("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3")
            if (OP == 1) CH1("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3")
            if (OP == 5) CH1("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8")
            if (OP == 6) CH1("v_mad_u64_u32 %4, vcc, %0, %8, %4\n v_mad_u64_u32 %5, vcc, %1, %8, %5\n v_mad_u64_u32 %6, vcc, %2, %8, %6\n v_mad_u64_u32 %7, vcc, %3, %8, %7")
            if (OP == 7) CH1("v_mad_i64_i32 %4, vcc, %0, %8, %4\n v_mad_i64_i32 %5, vcc, %1, %8, %5\n v_mad_i64_i32 %6, vcc, %2, %8, %6\n v_mad_i64_i32 %7, vcc, %3, %8, %7")
            if (OP == 8) CH1("v_ashrrev_i64 %4, 1, %4\n v_ashrrev_i64 %5, 1, %5\n v_ashrrev_i64 %6, 1, %6\n v_ashrrev_i64 %7, 1, %7")
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + (u32)(A + B + C + D);
}
static int g_scale = 1;
template <int OP, int WIDE> void run(const char* name, u32* out, int ncu, int waves_per_simd) {
    const int per_rept = WIDE ? 4 : (OP == 12 ? 2 : 1);
    const int iters = g_scale * 512 / per_rept;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_rate<OP, WIDE>), dim3(ncu * 4 * waves_per_simd), dim3(64), 0, 0, out, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    const double instr_per_wave = (double)iters * N * per_rept;
    printf("%-34s %s waves/SIMD=%d  %7.3f ms  %5.2f cycles per instruction per wave, %5.2f per SIMD issue slot (2.4 GHz)\n", name, WIDE ? "4 chains" : "1 chain ", waves_per_simd, best,
           best * 1e-3 * 2.4e9 / instr_per_wave, best * 1e-3 * 2.4e9 / instr_per_wave / waves_per_simd);
}
int main(int argc, char** argv) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    u32* out; hipMalloc(&out, sizeof(u32) * p.multiProcessorCount * 64 * 64);
    const int n = p.multiProcessorCount;
    if (argc > 1) {                                                        // sustained: ubench_rate <scale>  (scale x 1 ms per line)
        g_scale = atoi(argv[1]);
        for (int rep = 0; rep < 3; rep++) { run<7, 1>("v_mad_i64_i32 sustained", out, n, 2); run<1, 1>("v_add_u32 sustained", out, n, 2); }
        return 0;
    }
    for (int w : {1, 2, 4}) {
        run<0, 0>("v_add_u32 literal", out, n, w); run<1, 0>("v_add_u32", out, n, w); run<2, 0>("v_and_b32", out, n, w); run<3, 0>("v_lshrrev_b32", out, n, w);
        run<4, 0>("v_add3_u32", out, n, w); run<5, 0>("v_mul_lo_u32", out, n, w); run<6, 0>("v_mad_u64_u32", out, n, w); run<7, 0>("v_mad_i64_i32", out, n, w);
        run<8, 0>("v_ashrrev_i64", out, n, w); run<9, 0>("v_mov_b32_dpp quad_perm", out, n, w); run<10, 0>("v_bfi_b32", out, n, w); run<11, 0>("v_sub_u32", out, n, w);
        run<12, 0>("v_mov_b32 pair", out, n, w); run<13, 0>("v_and_or_b32", out, n, w); run<14, 0>("v_alignbit_b32", out, n, w); run<15, 0>("v_lshl_add_u32", out, n, w);
        run<0, 1>("v_add_u32 literal", out, n, w); run<1, 1>("v_add_u32", out, n, w); run<5, 1>("v_mul_lo_u32", out, n, w); run<6, 1>("v_mad_u64_u32", out, n, w);
        run<7, 1>("v_mad_i64_i32", out, n, w); run<8, 1>("v_ashrrev_i64", out, n, w);
    }
    return 0;
}
