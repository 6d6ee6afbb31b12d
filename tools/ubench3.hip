// Does the carry-out SGPR of v_mad_u64_u32 serialise back-to-back MADs when a wave is alone on its SIMD?
// A: every MAD writes vcc; B: the (unused) carry-out rotates over four SGPR pairs; C: v_mul_lo_u32 for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64; typedef unsigned int u32;
template <int MODE> __global__ void __launch_bounds__(64) k(u32* out, int iters, u32 seed) {
  u32 a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
  u64 acc[8]; u32 r[8];
  for (int i = 0; i < 8; i++) { acc[i] = a + i; r[i] = a * 7 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      } else if (MODE == 1) {
        asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[22:23], %8, %9, %1\n v_mad_u64_u32 %2, s[24:25], %8, %9, %2\n v_mad_u64_u32 %3, s[26:27], %8, %9, %3\n"
                     "v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[22:23], %8, %9, %5\n v_mad_u64_u32 %6, s[24:25], %8, %9, %6\n v_mad_u64_u32 %7, s[26:27], %8, %9, %7\n"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) : "v"(a), "v"(b)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
      } else if (MODE == 3) {       // one dependent chain, vcc
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
      } else if (MODE == 4) {       // one dependent chain, rotating sdst
        asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n v_mad_u64_u32 %0, s[22:23], %1, %2, %0\n v_mad_u64_u32 %0, s[24:25], %1, %2, %0\n v_mad_u64_u32 %0, s[26:27], %1, %2, %0\n"
                     "v_mad_u64_u32 %0, s[20:21], %1, %2, %0\n v_mad_u64_u32 %0, s[22:23], %1, %2, %0\n v_mad_u64_u32 %0, s[24:25], %1, %2, %0\n v_mad_u64_u32 %0, s[26:27], %1, %2, %0\n"
                     : "+v"(acc[0]) : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
      } else if (MODE == 5) {       // v_mad_i64_i32 signed, 8 chains, vcc
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      }
    }
  }
  u32 s = 0;
  for (int i = 0; i < 8; i++) s += (u32)acc[i] + (u32)(acc[i] >> 32) + r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> float run(u32* out, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1u); hipDeviceSynchronize();
  float best = 1e9;
  for (int i = 0; i < 3; i++) { hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 1u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}
int main() {
  u32* out; hipMalloc(&out, 4 * 64 * 4096);
  const int iters = 20000;
  const char* names[] = {"mad_u64 x8 chains, vcc", "mad_u64 x8 chains, rotating sdst", "mul_lo x8 chains", "mad_u64 1 chain, vcc", "mad_u64 1 chain, rotating sdst", "mad_i64 x8 chains, vcc"};
  for (int blocks : {1, 1024, 2048}) {        // 1 wave on the chip; 1 per SIMD; 2 per SIMD
    float ms[6] = {run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters), run<3>(out, blocks, iters), run<4>(out, blocks, iters), run<5>(out, blocks, iters)};
    for (int m = 0; m < 6; m++) printf("blocks=%4d %-34s %8.3f ms  %.2f ns/instr/wave\n", blocks, names[m], ms[m], ms[m] * 1e6 / (iters * 32.0));
  }
  return 0;
}
