// the multiply-add COUNT of an Fq2 product in the lane-pair layout with 14 x 28-bit limbs (R = 2^392: 3 x 196 MACs)
// against the library's 15 x 27-bit limbs (R = 2^405: 3 x 225), timed as the kernels run them -- two waves per SIMD, dependent chain,
// operands prepared through DPP exactly as fp2_pair.inc does -- together with what 28-bit limbs force in exchange: only 11 bits of
// value head-room (R / q = 2560 against 2^24), i.e. a value reduction wherever the tower code today just normalises limbs.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o tools/ubench_core28 tools/ubench_core28.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int32_t i32; typedef int64_t i64; typedef uint32_t u32;
template <int NL> struct Q;
template <> struct Q<15> { static constexpr int LB = 27; static constexpr i32 q[15] = {0x7ffaaab, 0x7dfffff, 0x7fffee7, 0x7ff58a9, 0x241eabf, 0x1ed61ec, 0x1cc34a8, 0x42895fb, 0x74b84f3, 0x59aec8e, 0x6d90d2e, 0x5258dd3, 0x397fe69, 0x40223d4, 0x6}; static constexpr u32 qinv = 0x7fcfffd; };
template <> struct Q<14> { static constexpr int LB = 28; static constexpr i32 q[14] = {0xfffaaab, 0xfefffff, 0x3ffffb9, 0xfffeb15, 0x6241eab, 0xa0f6b0f, 0xf6730d2, 0xf38512b, 0x4774b84, 0x4bacd76, 0xba7b643, 0xe69a4b1, 0x1ea397f, 0x1a011}; static constexpr u32 qinv = 0xffcfffd; };
__device__ __forceinline__ i32 dpp_swap(i32 x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ i32 dpp_even(i32 x) { return __builtin_amdgcn_update_dpp(0, x, 0xA0, 0xf, 0xf, true); }
__device__ __forceinline__ i32 bfi(i32 m, i32 a, i32 b) { i32 r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b)); return r; }
template <int NL> struct V { i32 v[NL]; };
// the lane-pair Fq2 product (fp2_pair.inc: fp2p_mul_body): even lane a0 b0 - a1 b1, odd lane a1 b0 + a0 b1, one fused Montgomery pass
template <int NL>
__device__ __noinline__ V<NL> mul2(V<NL> am, V<NL> bm) {
    constexpr int LB = Q<NL>::LB; constexpr i32 MASK = (1 << LB) - 1;
    const i32 odd = -(i32)(threadIdx.x & 1);
    i32 ao[NL], p[NL], q[NL], m[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { ao[i] = dpp_swap(am.v[i]); p[i] = dpp_even(bm.v[i]); q[i] = bfi(odd, bm.v[i], 0 - dpp_swap(bm.v[i])); }
    V<NL> r; i64 acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc += (i64)am.v[i] * p[k - i]; acc += (i64)ao[i] * q[k - i]; }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)m[i] * Q<NL>::q[k - i];
        m[k] = (i32)((u32)(i32)acc * Q<NL>::qinv) & MASK;
        acc += (i64)m[k] * Q<NL>::q[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc += (i64)am.v[i] * p[k - i]; acc += (i64)ao[i] * q[k - i]; }
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (i64)m[i] * Q<NL>::q[k - i];
        r.v[k - NL] = (i32)acc & MASK;
        acc >>= LB;
    }
    r.v[NL - 1] = (i32)acc;
    return r;
}
// the lane-pair Fq2 square (fp2p_sqr_body): one single-product pass per lane
template <int NL>
__device__ __noinline__ V<NL> sqr2(V<NL> am) {
    constexpr int LB = Q<NL>::LB; constexpr i32 MASK = (1 << LB) - 1;
    const i32 odd = -(i32)(threadIdx.x & 1);
    i32 x[NL], y[NL], m[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { const i32 ao = dpp_swap(am.v[i]); x[i] = am.v[i] + bfi(odd, am.v[i], ao); y[i] = bfi(odd, ao, am.v[i] - ao); }
    V<NL> r; i64 acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (i64)x[i] * y[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)m[i] * Q<NL>::q[k - i];
        m[k] = (i32)((u32)(i32)acc * Q<NL>::qinv) & MASK;
        acc += (i64)m[k] * Q<NL>::q[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (i64)x[i] * y[k - i];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (i64)m[i] * Q<NL>::q[k - i];
        r.v[k - NL] = (i32)acc & MASK;
        acc >>= LB;
    }
    r.v[NL - 1] = (i32)acc;
    return r;
}
// value reduction (fp.cuh: fp_reduce): subtract round(value / q) q with exact carries; what 28-bit limbs need at every stored result
template <int NL>
__device__ __forceinline__ V<NL> reduce(V<NL> y) {
    constexpr int LB = Q<NL>::LB; constexpr i32 MASK = (1 << LB) - 1;
    const float top = (float)y.v[NL - 1] * (float)(1 << LB) + (float)y.v[NL - 2];
    const float qtop = (float)Q<NL>::q[NL - 1] * (float)(1 << LB) + (float)Q<NL>::q[NL - 2];
    const i32 k = (i32)floorf(top * (1.0f / qtop));
    V<NL> r; i64 c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) { c += (i64)y.v[i] - (i64)k * Q<NL>::q[i]; r.v[i] = (i32)c & MASK; c >>= LB; }
    c += (i64)y.v[NL - 1] - (i64)k * Q<NL>::q[NL - 1];
    r.v[NL - 1] = (i32)c;
    return r;
}
// carry-free limb normalisation (fp.cuh: fp_norm): what 27-bit limbs do at the same places
template <int NL>
__device__ __forceinline__ V<NL> norm(V<NL> x) {
    constexpr int LB = Q<NL>::LB; constexpr i32 MASK = (1 << LB) - 1;
    V<NL> r; r.v[0] = x.v[0] & MASK;
#pragma unroll
    for (int i = 1; i < NL - 1; i++) r.v[i] = (x.v[i] & MASK) + (x.v[i - 1] >> LB);
    r.v[NL - 1] = x.v[NL - 1] + (x.v[NL - 2] >> LB);
    return r;
}
// MODE 0: chain of products; 1: chain of squares; 2: a Granger-Scott-shaped step -- per iteration two products (one Fq4 squaring by
// products) whose two outputs are 3 t -+ 2 z combinations brought back to storage form: norm (27-bit) or reduce (28-bit)
template <int NL, int MODE>
__global__ void __launch_bounds__(64, 2) k_chain(i32* io, int iters) {
    V<NL> x, y;
    for (int i = 0; i < NL; i++) { x.v[i] = io[(threadIdx.x + 64 * blockIdx.x) * 32 + i] & ((1 << Q<NL>::LB) - 1); y.v[i] = x.v[i] ^ 0x155555; }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) x = mul2<NL>(x, y);
        else if (MODE == 1) x = sqr2<NL>(x);
        else {
            V<NL> s, t;
#pragma unroll
            for (int i = 0; i < NL; i++) { s.v[i] = x.v[i] + y.v[i]; t.v[i] = x.v[i] + 2 * y.v[i]; }
            const V<NL> v = mul2<NL>(x, y), w = mul2<NL>(s, t);
            V<NL> a, b;
#pragma unroll
            for (int i = 0; i < NL; i++) { a.v[i] = 3 * (w.v[i] - v.v[i]) - 2 * x.v[i]; b.v[i] = 6 * v.v[i] + 2 * y.v[i]; }
            if (NL == 14) { x = reduce<NL>(a); y = reduce<NL>(b); } else { x = norm<NL>(a); y = norm<NL>(b); }
        }
    }
    for (int i = 0; i < NL; i++) io[(threadIdx.x + 64 * blockIdx.x) * 32 + i] = x.v[i] + y.v[0];
}
template <int NL, int MODE> static double run(i32* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<NL, MODE>), dim3(2048), dim3(64), 0, 0, d, 16);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL((k_chain<NL, MODE>), dim3(2048), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    return best;
}
int main() {
    i32* d; hipMalloc(&d, 2048 * 64 * 32 * 4);
    std::vector<i32> h(2048 * 64 * 32); for (size_t i = 0; i < h.size(); i++) h[i] = (i32)(i * 2654435761u);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int it = 4000;
    const double m15 = run<15, 0>(d, it), m14 = run<14, 0>(d, it), s15 = run<15, 1>(d, it), s14 = run<14, 1>(d, it), g15 = run<15, 2>(d, it / 2), g14 = run<14, 2>(d, it / 2);
    // 2 048 waves = two per SIMD on 1 024 SIMDs: ns per wave-level call = ms * 1e6 / iters; both waves of a SIMD run concurrently
    printf("lane-pair Fq2 PRODUCT  15x27: %.1f ns per call per wave   14x28: %.1f ns   ratio %.3f\n", m15 * 1e6 / it, m14 * 1e6 / it, m14 / m15);
    printf("lane-pair Fq2 SQUARE   15x27: %.1f ns                     14x28: %.1f ns   ratio %.3f\n", s15 * 1e6 / it, s14 * 1e6 / it, s14 / s15);
    printf("Fq4-squaring-shaped step (2 products + recombination + storage form): 15x27 with norm %.1f ns   14x28 with value reduction %.1f ns   ratio %.3f\n",
           g15 * 1e6 / (it / 2), g14 * 1e6 / (it / 2), g14 / g15);
    return 0;
}
