// Issue rate of the library's own multiply cores in isolation (no memory traffic inside the timed loop), two waves per SIMD:
// how far the pairing kernels' 5.1-5.5 cycles per instruction are from what their dominant code does alone.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I bls_amd/csrc -o tools/ubench_core tools/ubench_core.hip
#include "pairing.cuh"
#include "device_io.cuh"
#include <cstdio>
namespace P2 = blsmi::pairl;
template <int MODE> __global__ void __launch_bounds__(64, 2) k_core(i32* out, int iters, int seed) {
    P2::Fp2S a, b;
    for (int i = 0; i < NL; i++) { a.c.v[i] = (seed * 7 + i * 131 + threadIdx.x * 17) & MASK; b.c.v[i] = (seed * 3 + i * 71 + threadIdx.x) & MASK; }
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) { a = P2::fp2_store(P2::fp2_mul(a, b)); }
        if (MODE == 1) { a = P2::fp2_store(P2::fp2_sqr(a)); }
        if (MODE == 2) { const auto t = P2::fp2_mul(a, b); b = P2::fp2_store(P2::fp2_add(P2::fp2_sqr(a), t)); a = P2::fp2_store(P2::fp2_sub(t, b)); }
    }
    if (MODE == 6 || MODE == 7 || MODE == 8) {
        // what one trip of 45 words through the lane's scratch costs beside a core call: the callee's entry s_waitcnt drains it
        volatile i32 priv[64];
        for (int i = 0; i < 45; i++) priv[i] = a.c.v[i % NL] + i;
        for (int it = 0; it < iters; it++) {
            if (MODE == 6 || MODE == 8) for (int i = 0; i < 45; i++) priv[i] = a.c.v[i % NL] ^ it;      // stores, then the call
            if (MODE == 7 || MODE == 8) { i32 acc = 0; for (int i = 0; i < 45; i++) acc += priv[i]; b.c.v[0] = (b.c.v[0] + acc) & MASK; }   // loads, then the call
            a = P2::fp2_store(P2::fp2_mul(a, b));
        }
    }
    if (MODE == 3 || MODE == 4 || MODE == 5) {
        P2::Fp12S f = P2::fp12_one();
        f.c0.c1 = a; f.c1.c2 = b; f.c1.c0 = a;
        P2::G2Proj r; r.x = a; r.y = b; r.z = P2::fp2_one();
        P2::Fp2S o0 = a, o1 = b, o2 = a;
        const FpS px = a.c, py = b.c;
        for (int it = 0; it < iters; it++) {
            if (MODE == 3) P2::ell_sqr(f, o0, o1, o2, px, py);
            if (MODE == 4) P2::doubling_step_h(r, o0, o1, o2);
            if (MODE == 5) { P2::doubling_step_h(r, o0, o1, o2); P2::ell_sqr(f, o0, o1, o2, px, py); }
        }
        a = P2::fp2_store(P2::fp2_add(P2::fp2_add(f.c0.c0, f.c1.c1), P2::fp2_add(r.x, o1)));
    }
    for (int i = 0; i < NL; i++) out[(blockIdx.x * 64 + threadIdx.x) * NL + i] = a.c.v[i] + b.c.v[i];
}
template <int MODE> void run(const char* name, i32* out, int ncu, double instr_per_iter) {
    const int iters = MODE >= 3 ? 200 : 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_core<MODE>, dim3(ncu * 8), dim3(64), 0, 0, out, iters, rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    printf("%-30s %7.3f ms  %6.0f cycles per iteration per wave; at %4.0f instructions per iteration: %.2f cycles per instruction per SIMD issue slot\n", name, best, best * 1e-3 * 2.4e9 / iters,
           instr_per_iter, best * 1e-3 * 2.4e9 / iters / instr_per_iter / 2);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    i32* out; hipMalloc(&out, sizeof(i32) * p.multiProcessorCount * 8 * 64 * NL);
    run<0>("fp2 mul (fp2p_mul_core 874)", out, p.multiProcessorCount, 874 + 50);
    run<1>("fp2 sqr (fp2p_sqr_core 680)", out, p.multiProcessorCount, 680 + 50);
    run<2>("mul + sqr + add + sub", out, p.multiProcessorCount, 874 + 680 + 130);
    // cores alone (measured above): fp2 mul 3931, fp2 sqr 2905, fp mul ~2560 cycles per SIMD slot
    run<3>("ell_sqr (25 mul + 2 fp mul)", out, p.multiProcessorCount, 3547 + 25 * 874 + 2 * 596);
    run<4>("doubling_step_h (4 mul + 5 sqr)", out, p.multiProcessorCount, 1013 + 4 * 874 + 5 * 680);
    run<6>("45 scratch stores + fp2 mul", out, p.multiProcessorCount, 924 + 90);
    run<7>("45 scratch loads + fp2 mul", out, p.multiProcessorCount, 924 + 90);
    run<8>("stores + loads + fp2 mul", out, p.multiProcessorCount, 924 + 180);
    run<5>("one Miller step (both)", out, p.multiProcessorCount, 3547 + 25 * 874 + 2 * 596 + 1013 + 4 * 874 + 5 * 680);
    return 0;
}
