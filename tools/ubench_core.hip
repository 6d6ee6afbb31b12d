// Issue rate of the library's own multiply cores in isolation (no memory traffic inside the timed loop), two waves per SIMD:
// how far the pairing kernels' 5.1-5.5 cycles per instruction are from what their dominant code does alone.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I bls_amd/csrc -o tools/ubench_core tools/ubench_core.hip
#include "pairing.cuh"
#include "device_io.cuh"
#include <cstdio>
namespace P2 = blsmi::pairl;
#define AS5 __attribute__((address_space(5)))
typedef int v4i __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ T fetch5(const AS5 T* src) {
    T dst; const AS5 int* p = (const AS5 int*)src; int* q = (int*)&dst;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) q[i] = p[i];
    return dst;
}
template <class T> __device__ __forceinline__ void store5(AS5 T* dst, const T& src) {
    const int* p = (const int*)&src; AS5 int* q = (AS5 int*)dst;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) q[i] = p[i];
}
__device__ long long g_ts[8];
__device__ __noinline__ void dbl_step_p(AS5 P2::G2Proj* r, AS5 P2::Fp2S* o) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    P2::G2Proj rr = fetch5(r);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    P2::Fp2S a0, a1, a2;
    P2::doubling_step_h_i(rr, a0, a1, a2);
    __builtin_amdgcn_sched_barrier(0);
    const long long t2 = __builtin_amdgcn_s_memtime();
    store5(r, rr); store5(o, a0); store5(o + 1, a1); store5(o + 2, a2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t3 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_ts[0] = t0; g_ts[1] = t1; g_ts[2] = t2; g_ts[3] = t3; }
}
// ell_sqr (f <- (f * line)^2, pairing_body.inc) as PHASES: every value that must wait in memory is stored, and everything the next
// multiplications need is fetched, at a few points where ONE memory round trip covers both; between those points the live set stays
// below what survives an out-of-line core call (256 registers - the cores' 107), so the compiler has nothing to spill.
namespace blsmi { namespace pairl {
struct Pad { Fp6S bb, t, ab, x, y; };
__device__ __noinline__ void ell_sqr_phased(AS5 Fp12S* fp, const AS5 Fp2S* o, const AS5 FpS* pxy, AS5 Pad* pad) {
    // round trip 1: the line, the point, f.c1
    const Fp2S o0 = fetch5(o), o1 = fetch5(o + 1), l0 = fetch5(o + 2);
    const FpS px = fetch5(pxy), py = fetch5(pxy + 1);
    const Fp6S fc1 = fetch5(&fp->c1);
    __builtin_amdgcn_sched_barrier(0);
    const Fp2S l4 = fp2_store(fp2_mul_fp(o0, py)), l1 = fp2_store(fp2_mul_fp(o1, px));
    const Fp6S bb = fp6_store(fp6_mul_by_1(fc1, l4));
    __builtin_amdgcn_sched_barrier(0);
    // 2: bb out, f.c0 in
    store5(&pad->bb, bb);
    const Fp6S fc0 = fetch5(&fp->c0);
    __builtin_amdgcn_sched_barrier(0);
    const Fp6S s = fp6_store(fp6_add(fc0, fc1));
    const Fp6S t = fp6_store(fp6_mul_by_01(s, l0, fp2_add(l1, l4)));
    __builtin_amdgcn_sched_barrier(0);
    // 3: t out, f.c0 in again
    store5(&pad->t, t);
    const Fp6S fc0b = fetch5(&fp->c0);
    __builtin_amdgcn_sched_barrier(0);
    const Fp6S aa = fp6_store(fp6_mul_by_01(fc0b, l0, l1));
    __builtin_amdgcn_sched_barrier(0);
    // 4: bb, t in
    const Fp6S bb2 = fetch5(&pad->bb), t2 = fetch5(&pad->t);
    __builtin_amdgcn_sched_barrier(0);
    const Fp6S g0 = fp6_store(fp6_add(fp6_mul_nr(bb2), aa)), g1 = fp6_store(fp6_sub(fp6_sub(t2, aa), bb2));
    const Fp6S x = fp6_store(fp6_add(fp6_mul_nr(g1), g0)), y = fp6_store(fp6_add(g0, g1));
    // 5: x, y out (drained at the next core call)
    store5(&pad->x, x); store5(&pad->y, y);
    __builtin_amdgcn_sched_barrier(0);
    const Fp6S ab = fp6_store(fp6_mul(g0, g1));
    __builtin_amdgcn_sched_barrier(0);
    // 6: ab out; x, y in
    store5(&pad->ab, ab);
    const Fp6S x2 = fetch5(&pad->x), y2 = fetch5(&pad->y);
    __builtin_amdgcn_sched_barrier(0);
    const Fp6S tt = fp6_store(fp6_mul(x2, y2));
    __builtin_amdgcn_sched_barrier(0);
    // 7: ab in
    const Fp6S ab2 = fetch5(&pad->ab);
    __builtin_amdgcn_sched_barrier(0);
    Fp12S r;
    r.c0 = fp6_store(fp6_sub(fp6_sub(tt, ab2), fp6_mul_nr(ab2)));
    r.c1 = fp6_store(fp6_add(ab2, ab2));
    store5(fp, r);
}
} }
using P2::Pad; using P2::ell_sqr_phased;
// An Fq12 multiplication on references as FOUR memory events (what waits is stored and what the next six products need is fetched at
// the same points), nothing for the compiler to spill in between: the out-of-line version the library used spends more cycles on
// scattered spills -- each drained at the next core call -- than on its 18 products.
namespace blsmi { namespace pairl {
template <class T> __device__ __forceinline__ T fetch_g(const T* src) {
    T dst; const int* p = (const int*)src; int* q = (int*)&dst;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) q[i] = p[i];
    return dst;
}
template <class T> __device__ __forceinline__ void store_g(T* dst, const T& src) {
    const int* p = (const int*)&src; int* q = (int*)dst;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++) q[i] = p[i];
}
__device__ __noinline__ void fp12_mul_phased(Fp12S* r, const Fp12S* a, const Fp12S* b) {
    Fp6S pad_[2];
    Fp6S* pad = pad_;
    asm volatile("" : "+v"(pad));                                         // the two waiting products stay in memory
    // 1: a.c0, b.c0
    Fp6S x = fetch_g(&a->c0), y = fetch_g(&b->c0);
    __builtin_amdgcn_sched_barrier(0);
    const Fp6S t0 = fp6_store(fp6_mul(x, y));
    __builtin_amdgcn_sched_barrier(0);
    // 2: t0 out; a.c1, b.c1 in
    store_g(&pad[0], t0);
    x = fetch_g(&a->c1); y = fetch_g(&b->c1);
    __builtin_amdgcn_sched_barrier(0);
    const Fp6S t1 = fp6_store(fp6_mul(x, y));
    __builtin_amdgcn_sched_barrier(0);
    // 3: t1 out; a.c0, b.c0 in again for the sums
    store_g(&pad[1], t1);
    const Fp6S sa = fp6_store(fp6_add(fetch_g(&a->c0), x)), sb = fp6_store(fp6_add(fetch_g(&b->c0), y));
    __builtin_amdgcn_sched_barrier(0);
    const Fp6S t2 = fp6_store(fp6_mul(sa, sb));
    __builtin_amdgcn_sched_barrier(0);
    // 4: t0, t1 in; the result out
    const Fp6S u0 = fetch_g(&pad[0]), u1 = fetch_g(&pad[1]);
    __builtin_amdgcn_sched_barrier(0);
    Fp12S res;
    res.c0 = fp6_store(fp6_add(fp6_mul_nr(u1), u0));
    res.c1 = fp6_store(fp6_sub(fp6_sub(t2, u0), u1));
    store_g(r, res);
}
} }
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
    if (MODE >= 12 && MODE <= 17) {
        // the final exponentiation's squarings: plain Granger-Scott (9 square cores), compressed (6), and one exp_by_x
        P2::Fp12S f = P2::fp12_one();
        f.c0.c1 = a; f.c1.c2 = b; f.c1.c0 = a; f.c0.c2 = b; f.c1.c1 = a;
        for (int it = 0; it < iters; it++) {
            if (MODE == 12) for (int k = 0; k < 16; k++) f = P2::fp12_cyclotomic_sqr(f);
            if (MODE == 13) { P2::CycCompressed c = P2::cyc_compress(f); for (int k = 0; k < 16; k++) c = P2::cyc_compressed_sqr(c); f.c0.c1 = c.z4; f.c0.c2 = c.z3; f.c1.c0 = c.z2; f.c1.c2 = c.z5; }
            if (MODE == 14) { P2::Fp12S o; P2::exp_by_x(o, f, BLSMI_X_ABS); f = o; }
            if (MODE == 15) { P2::Fp12S g = f; for (int k = 0; k < 4; k++) { P2::Fp12S t = f; P2::nf_fp12_mul(t, t, g); f = t; } }
            if (MODE == 16) f = P2::cyc_sqr_run(f, 16);
            if (MODE == 17) { P2::Fp12S g = f; for (int k = 0; k < 4; k++) { P2::Fp12S t = f; P2::fp12_mul_phased(&t, &t, &g); f = t; } }
        }
        a = P2::fp2_store(P2::fp2_add(P2::fp2_add(f.c0.c0, f.c1.c1), P2::fp2_add(f.c0.c1, f.c1.c2)));
    }
    if (MODE == 3 || MODE == 4 || MODE == 5 || MODE == 9 || MODE == 10 || MODE == 11) {
        P2::Fp12S f = P2::fp12_one();
        f.c0.c1 = a; f.c1.c2 = b; f.c1.c0 = a;
        P2::G2Proj r; r.x = a; r.y = b; r.z = P2::fp2_one();
        P2::Fp2S o0 = a, o1 = b, o2 = a;
        const FpS px = a.c, py = b.c;
        for (int it = 0; it < iters; it++) {
            if (MODE == 3) P2::ell_sqr(f, o0, o1, o2, px, py);
            if (MODE == 11) { P2::Fp2S oo[3] = {o0, o1, o2}; FpS pp[2] = {px, py}; Pad pad; ell_sqr_phased((AS5 P2::Fp12S*)&f, (const AS5 P2::Fp2S*)oo, (const AS5 FpS*)pp, (AS5 Pad*)&pad); }
            if (MODE == 4) P2::doubling_step_h(r, o0, o1, o2);
            if (MODE == 9) P2::doubling_step_h_i(r, o0, o1, o2);
            if (MODE == 10) { P2::Fp2S oo[3]; dbl_step_p((AS5 P2::G2Proj*)&r, (AS5 P2::Fp2S*)oo); o0 = oo[0]; o1 = oo[1]; o2 = oo[2]; }
            if (MODE == 5) { P2::doubling_step_h(r, o0, o1, o2); P2::ell_sqr(f, o0, o1, o2, px, py); }
        }
        a = P2::fp2_store(P2::fp2_add(P2::fp2_add(f.c0.c0, f.c1.c1), P2::fp2_add(r.x, o1)));
    }
    for (int i = 0; i < NL; i++) out[(blockIdx.x * 64 + threadIdx.x) * NL + i] = a.c.v[i] + b.c.v[i];
}
template <int MODE> void run(const char* name, i32* out, int ncu, double instr_per_iter) {
    const int iters = MODE == 14 ? 8 : MODE >= 12 ? 40 : MODE >= 3 ? 200 : 2000;
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
    run<11>("ell_sqr in phases", out, p.multiProcessorCount, 3547 + 25 * 874 + 2 * 596);
    run<4>("doubling_step_h (4 mul + 5 sqr)", out, p.multiProcessorCount, 1013 + 4 * 874 + 5 * 680);
    run<9>("doubling_step_h inlined (registers)", out, p.multiProcessorCount, 1013 + 4 * 874 + 5 * 680);
    run<10>("doubling_step_h, scratch pointers", out, p.multiProcessorCount, 1013 + 4 * 874 + 5 * 680);
    { long long ts[8]; hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof ts);
      printf("   timeline of one call (s_memtime ticks, 100 MHz): loads %lld, compute %lld, stores+drain %lld\n", ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2]); }
    run<12>("16 plain cyclotomic squarings", out, p.multiProcessorCount, 16 * (9 * 680 + 550));
    run<13>("16 compressed squarings", out, p.multiProcessorCount, 16 * (6 * 680 + 360));
    run<14>("exp_by_x", out, p.multiProcessorCount, 15 * 6660 + 48 * 4440 + 2 * 28000 + 5 * 18600);
    run<15>("4 nf_fp12_mul (18 mul cores each)", out, p.multiProcessorCount, 4 * 18600);
    run<17>("4 Fq12 multiplications in four memory events", out, p.multiProcessorCount, 4 * 18600);
    run<16>("cyc_sqr_run(16): 16 compressed + decompression", out, p.multiProcessorCount, 16 * 4440 + 28000);
    run<6>("45 scratch stores + fp2 mul", out, p.multiProcessorCount, 924 + 90);
    run<7>("45 scratch loads + fp2 mul", out, p.multiProcessorCount, 924 + 90);
    run<8>("stores + loads + fp2 mul", out, p.multiProcessorCount, 924 + 180);
    run<5>("one Miller step (both)", out, p.multiProcessorCount, 3547 + 25 * 874 + 2 * 596 + 1013 + 4 * 874 + 5 * 680);
    return 0;
}
