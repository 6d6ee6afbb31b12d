// blsmi.hip -- kernels and the C ABI (include/blsmi.h) of libblsmi.so.
//
// This translation unit holds the one-tuple-per-lane kernels (hash-to-curve, scalar multiplication, sums, MSM, wire
// format, and the pairing kernels selectable with BLSMI_LAYOUT=single) and the host side; the default pairing kernels
// (lane pair per tuple, two waves per SIMD) are in pair_kernels.inc, the verify-path kernels in verify_kernels.inc,
// the bucket-method MSM in msm.inc, the verify-path host code in verify_host.inc.  64-lane workgroups throughout.
// A one-tuple-per-lane pairing keeps f (180 words), R (90) and P, Q (90) per lane and wants the whole 512-entry
// register file, i.e. one wave per SIMD; G1-side kernels fit 256 registers and run two.
#include "../../include/blsmi.h"
#include "pairing.cuh"
#include "hash.cuh"
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>

using namespace blsmi;

#define WG 64
#ifndef BLSMI_WAVES_PER_SIMD
#define BLSMI_WAVES_PER_SIMD 1
#endif
#define KERNEL __global__ void __launch_bounds__(WG, BLSMI_WAVES_PER_SIMD)
// G1-side kernels keep far less live state (Fq, not Fq2/Fq12): 256 registers, two waves per SIMD
#define KERNEL2 __global__ void __launch_bounds__(WG, 2)

// ------------------------------------------------------------------------------------------------
// device-side I/O helpers
// ------------------------------------------------------------------------------------------------
// internal structure-of-arrays buffers: word (e, j) of tuple t lives at buf[(e*NL + j)*n + t]
BLSMI_DEV void soa_store(i32* buf, size_t n, size_t t, int e, const FpS& x) {
#pragma unroll
    for (int j = 0; j < NL; j++) buf[((size_t)e * NL + j) * n + t] = x.v[j];
}
BLSMI_DEV FpS soa_load(const i32* buf, size_t n, size_t t, int e) {
    FpS x;
#pragma unroll
    for (int j = 0; j < NL; j++) x.v[j] = buf[((size_t)e * NL + j) * n + t];
    return x;
}
BLSMI_DEV void soa_store12(i32* buf, size_t n, size_t t, const Fp12S& f) {
    const FpS* c = reinterpret_cast<const FpS*>(&f);
#pragma unroll
    for (int e = 0; e < 12; e++) soa_store(buf, n, t, e, c[e]);
}
BLSMI_DEV Fp12S soa_load12(const i32* buf, size_t n, size_t t) {
    Fp12S f;
    FpS* c = reinterpret_cast<FpS*>(&f);
#pragma unroll
    for (int e = 0; e < 12; e++) c[e] = soa_load(buf, n, t, e);
    return f;
}
// 48-byte big-endian field element at p (4-byte aligned) -> Montgomery
BLSMI_DEV FpS load_be48(const u8* p) {
    const u32* w32 = reinterpret_cast<const u32*>(p);
    u32 w[12];
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = __builtin_bswap32(w32[11 - j]);
    return fp_from_words(w);
}
template <int L, int V>
BLSMI_DEV void store_be48(u8* p, const Fp<L, V>& x) {
    u32 w[12];
    fp_to_words(x, w);
    u32* w32 = reinterpret_cast<u32*>(p);
#pragma unroll
    for (int j = 0; j < 12; j++) w32[11 - j] = __builtin_bswap32(w[j]);
}
BLSMI_DEV FpS load_m384(const u64* p) {
    const u32* w32 = reinterpret_cast<const u32*>(p);
    u32 w[12];
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = w32[j];
    return fp_from_mont384_words(w);
}
template <int L, int V>
BLSMI_DEV void store_m384(u64* p, const Fp<L, V>& x) {
    u32 w[12];
    fp_to_mont384_words(x, w);
    u32* w32 = reinterpret_cast<u32*>(p);
#pragma unroll
    for (int j = 0; j < 12; j++) w32[j] = w[j];
}
// ---- tuple I/O staged through LDS --------------------------------------------------------------------
// The host-facing records are array-of-structures (96 / 192 / 576 bytes per tuple).  A wave moves its
// 64 records between HBM and LDS with lane-contiguous dword accesses (256 B per wave instruction,
// fully coalesced) and each lane then works on its own record inside LDS.  Records are padded by one
// word in LDS so that the per-lane stride is odd (no bank conflicts on the per-lane side).
template <int WORDS>
BLSMI_DEV void tile_load(u32* lds, const u8* gbase, size_t first, size_t n) {
    const u32* g = reinterpret_cast<const u32*>(gbase) + first * WORDS;
    const size_t valid = (n - first < (size_t)WG ? n - first : (size_t)WG) * WORDS;
    for (int idx = threadIdx.x; idx < WG * WORDS; idx += WG)
        if ((size_t)idx < valid) lds[(idx / WORDS) * (WORDS + 1) + (idx % WORDS)] = g[idx];
    __syncthreads();
}
template <int WORDS>
BLSMI_DEV void tile_store(const u32* lds, u8* gbase, size_t first, size_t n) {
    __syncthreads();
    u32* g = reinterpret_cast<u32*>(gbase) + first * WORDS;
    const size_t valid = (n - first < (size_t)WG ? n - first : (size_t)WG) * WORDS;
    for (int idx = threadIdx.x; idx < WG * WORDS; idx += WG)
        if ((size_t)idx < valid) g[idx] = lds[(idx / WORDS) * (WORDS + 1) + (idx % WORDS)];
}
BLSMI_DEV FpS lds_be48(const u32* rec) {                                  // 12 big-endian words of this lane's record
    u32 w[12];
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = __builtin_bswap32(rec[11 - j]);
    return fp_from_words(w);
}
BLSMI_DEV G1Aff lds_g1(const u32* rec) { G1Aff a; a.x = lds_be48(rec); a.y = lds_be48(rec + 12); a.inf = 0; return a; }
BLSMI_DEV G2Aff lds_g2(const u32* rec) { G2Aff a; a.x.c0 = lds_be48(rec); a.x.c1 = lds_be48(rec + 12); a.y.c0 = lds_be48(rec + 24); a.y.c1 = lds_be48(rec + 36); a.inf = 0; return a; }

BLSMI_DEV G1Aff load_g1(const u8* p) { G1Aff a; a.x = load_be48(p); a.y = load_be48(p + 48); a.inf = 0; return a; }
BLSMI_DEV G2Aff load_g2(const u8* p) {
    G2Aff a; a.x.c0 = load_be48(p); a.x.c1 = load_be48(p + 48); a.y.c0 = load_be48(p + 96); a.y.c1 = load_be48(p + 144); a.inf = 0; return a;
}
BLSMI_DEV void store_g1(u8* p, const G1Aff& a) {
    if (a.inf) { u32* w = reinterpret_cast<u32*>(p); for (int i = 0; i < 24; i++) w[i] = 0; return; }
    store_be48(p, a.x); store_be48(p + 48, a.y);
}
BLSMI_DEV void store_g2(u8* p, const G2Aff& a) {
    if (a.inf) { u32* w = reinterpret_cast<u32*>(p); for (int i = 0; i < 48; i++) w[i] = 0; return; }
    store_be48(p, a.x.c0); store_be48(p + 48, a.x.c1); store_be48(p + 96, a.y.c0); store_be48(p + 144, a.y.c1);
}

// ------------------------------------------------------------------------------------------------
// kernels: pairing
// ------------------------------------------------------------------------------------------------
// Miller loop for one pair per tuple; f goes to the internal SoA buffer (or nowhere else).
KERNEL k_miller1(const u8* g1, const u8* g2, i32* fbuf, size_t n) {
    __shared__ u32 lds[WG * 49];
    const size_t first = (size_t)blockIdx.x * WG;
    const size_t t = first + threadIdx.x;
    const int rec = (t < n) ? (int)threadIdx.x : (int)(n - 1 - first);   // tail lanes redo the last tuple
    G1Aff p[1]; G2Aff q[1];
    tile_load<24>(lds, g1, first, n);
    p[0] = lds_g1(lds + rec * 25);
    __syncthreads();
    tile_load<48>(lds, g2, first, n);
    q[0] = lds_g2(lds + rec * 49);
    Fp12S f;
    miller_loop<1>(f, p, q);
    if (t < n) soa_store12(fbuf, n, t, f);
}
// mode 0: out = FE(f) as Montgomery-384 limbs; mode 1: out = f itself (no final exponentiation)
KERNEL k_final_exp(const i32* fbuf, u64* out, size_t n, int mode) {
    __shared__ u32 lds[WG * 145];
    const size_t first = (size_t)blockIdx.x * WG;
    const size_t t = first + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    Fp12S f = soa_load12(fbuf, n, tt);
    if (mode == 0) final_exponentiation(f);
    const FpS* c = reinterpret_cast<const FpS*>(&f);
    for (int e = 0; e < 12; e++) {                                       // this lane's 576-byte record, into LDS
        u32 w[12];
        fp_to_mont384_words(c[e], w);
#pragma unroll
        for (int j = 0; j < 12; j++) lds[threadIdx.x * 145 + 12 * e + j] = w[j];
    }
    tile_store<144>(lds, reinterpret_cast<u8*>(out), first, n);         // coalesced write-out
}
KERNEL k_fq12_from_m384(const u64* in, i32* fbuf, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    for (int e = 0; e < 12; e++) soa_store(fbuf, n, t, e, load_m384(in + 72 * t + 6 * e));
}

// ------------------------------------------------------------------------------------------------
// kernels: unit-level ops for the parity tests
// ------------------------------------------------------------------------------------------------
template <int W> struct Rec { FpS e[W]; };
template <int W> BLSMI_DEV Rec<W> rec_load(const u64* p, size_t t) { Rec<W> r; for (int i = 0; i < W; i++) r.e[i] = load_m384(p + (size_t)6 * (W * t + i)); return r; }
template <int W> BLSMI_DEV void rec_store(u64* p, size_t t, const Rec<W>& r) { for (int i = 0; i < W; i++) store_m384(p + (size_t)6 * (W * t + i), r.e[i]); }
template <class T, int W> BLSMI_DEV T& as(Rec<W>& r) { return *reinterpret_cast<T*>(&r); }

KERNEL k_debug_fq(int op, const u64* a, const u64* b, u64* out, u8* flag, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    const FpS x = load_m384(a + 6 * t);
    FpS y = fp_zero();
    if (op == BLSMI_OP_FQ_MUL || op == BLSMI_OP_FQ_ADD || op == BLSMI_OP_FQ_SUB) y = load_m384(b + 6 * t);
    FpS r = fp_zero();
    bool ok = true;
    switch (op) {
        case BLSMI_OP_FQ_MUL: r = fp_store(fp_mul(x, y)); break;
        case BLSMI_OP_FQ_SQR: r = fp_store(fp_sqr(x)); break;
        case BLSMI_OP_FQ_ADD: r = fp_store(fp_add(x, y)); break;
        case BLSMI_OP_FQ_SUB: r = fp_store(fp_sub(x, y)); break;
        case BLSMI_OP_FQ_NEG: r = fp_store(fp_neg(x)); break;
        case BLSMI_OP_FQ_INV: r = fp_inv(x); ok = !fp_is_zero(x); break;
        case BLSMI_OP_FQ_SQRT: r = fp_sqrt(x, ok); break;
    }
    store_m384(out + 6 * t, r);
    if (flag) flag[t] = ok ? 1 : 0;
}
KERNEL k_debug_fq2(int op, const u64* a, const u64* b, u64* out, u8* flag, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<2> ra = rec_load<2>(a, t), rb = ra, ro;
    if (op == BLSMI_OP_FQ2_MUL) rb = rec_load<2>(b, t);
    const Fp2S x = as<Fp2S>(ra), y = as<Fp2S>(rb);
    Fp2S r = fp2_zero();
    bool ok = true;
    switch (op) {
        case BLSMI_OP_FQ2_MUL: r = fp2_store(fp2_mul(x, y)); break;
        case BLSMI_OP_FQ2_SQR: r = fp2_store(fp2_sqr(x)); break;
        case BLSMI_OP_FQ2_INV: r = fp2_store(fp2_inv(x)); ok = !fp2_is_zero(x); break;
        case BLSMI_OP_FQ2_MUL_NR: r = fp2_store(fp2_mul_nr(x)); break;
        case BLSMI_OP_FQ2_SQRT: r = fp2_sqrt(x, ok); break;
        case BLSMI_OP_FQ2_SQRT_ANY: r = fp2_sqrt_any(x, ok); break;
    }
    as<Fp2S>(ro) = r;
    rec_store<2>(out, t, ro);
    if (flag) flag[t] = ok ? 1 : 0;
}
KERNEL k_debug_fq6(int op, const u64* a, const u64* b, u64* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<6> ra = rec_load<6>(a, t), rb = ra, ro;
    if (op == BLSMI_OP_FQ6_MUL) rb = rec_load<6>(b, t);
    const Fp6S x = as<Fp6S>(ra), y = as<Fp6S>(rb);
    Fp6S r = fp6_zero();
    switch (op) {
        case BLSMI_OP_FQ6_MUL: r = fp6_store(fp6_mul(x, y)); break;
        case BLSMI_OP_FQ6_SQR: r = fp6_store(fp6_sqr(x)); break;
        case BLSMI_OP_FQ6_INV: r = fp6_store(fp6_inv(x)); break;
        case BLSMI_OP_FQ6_FROB1: r = fp6_store(fp6_frob<1>(x)); break;
    }
    as<Fp6S>(ro) = r;
    rec_store<6>(out, t, ro);
}
KERNEL k_debug_fq12(int op, const u64* a, const u64* b, u64* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<12> ra = rec_load<12>(a, t), rb = ra, ro;
    if (op == BLSMI_OP_FQ12_MUL) rb = rec_load<12>(b, t);
    const Fp12S x = as<Fp12S>(ra), y = as<Fp12S>(rb);
    Fp12S r = fp12_one();
    switch (op) {
        case BLSMI_OP_FQ12_MUL: r = fp12_store(fp12_mul(x, y)); break;
        case BLSMI_OP_FQ12_SQR: r = fp12_store(fp12_sqr(x)); break;
        case BLSMI_OP_FQ12_INV: r = fp12_store(fp12_inv(x)); break;
        case BLSMI_OP_FQ12_FROB1: r = fp12_store(fp12_frob<1>(x)); break;
        case BLSMI_OP_FQ12_FROB2: r = fp12_store(fp12_frob<2>(x)); break;
        case BLSMI_OP_FQ12_FROB3: r = fp12_store(fp12_frob<3>(x)); break;
        case BLSMI_OP_FQ12_CYCLO_SQR: r = fp12_cyclotomic_sqr(x); break;
        case BLSMI_OP_FQ12_CYCLO_RUN16: r = cyc_sqr_run(x, 16); break;
    }
    as<Fp12S>(ro) = r;
    rec_store<12>(out, t, ro);
}
// Jacobian points as 3 (G1) / 6 (G2) Fq records x,y,z; infinity <=> z == 0 (g1.go:293)
template <class F, int W>
BLSMI_DEV void debug_curve(int dbl, const u64* a, const u64* b, u64* out, size_t t) {
    Rec<W> ra = rec_load<W>(a, t), rb = ra, ro;
    if (!dbl) rb = rec_load<W>(b, t);
    Jac<F> p, q, r;
    p.x = reinterpret_cast<F*>(&ra)[0]; p.y = reinterpret_cast<F*>(&ra)[1]; p.z = reinterpret_cast<F*>(&ra)[2]; p.inf = f_is_zero(p.z) ? -1 : 0;
    q.x = reinterpret_cast<F*>(&rb)[0]; q.y = reinterpret_cast<F*>(&rb)[1]; q.z = reinterpret_cast<F*>(&rb)[2]; q.inf = f_is_zero(q.z) ? -1 : 0;
    r = dbl ? jac_double(p) : jac_add(p, q);
    if (r.inf) r.z = field_consts<F>::zero();
    reinterpret_cast<F*>(&ro)[0] = r.x; reinterpret_cast<F*>(&ro)[1] = r.y; reinterpret_cast<F*>(&ro)[2] = r.z;
    rec_store<W>(out, t, ro);
}
KERNEL k_debug_curve(int op, const u64* a, const u64* b, u64* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    if (op == BLSMI_OP_G1_DOUBLE || op == BLSMI_OP_G1_ADD) debug_curve<FpS, 3>(op == BLSMI_OP_G1_DOUBLE, a, b, out, t);
    else debug_curve<Fp2S, 6>(op == BLSMI_OP_G2_DOUBLE, a, b, out, t);
}
// SWU helpers on a caller-chosen t (own kernels: the G1 helper keeps the two-waves-per-SIMD register budget of k_hash_g1)
KERNEL2 k_debug_swu_g1(const u64* a, u64* out, size_t n) {               // optimizedSWUMapHelper (g1.go:628-714)
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<3> r = rec_load<3>(a, t);
    G1Aff p; swu_g1_helper(p, reinterpret_cast<FpS*>(&r)[0]);
    reinterpret_cast<FpS*>(&r)[0] = p.x; reinterpret_cast<FpS*>(&r)[1] = p.y; reinterpret_cast<FpS*>(&r)[2] = fp_zero();
    rec_store<3>(out, t, r);
}
KERNEL k_debug_swu_g2(const u64* a, u64* out, size_t n) {                // OptimizedSWU2MapHelper (g2.go:933-1031)
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<6> r = rec_load<6>(a, t);
    G2Aff p; swu_g2_helper(p, reinterpret_cast<Fp2S*>(&r)[0]);
    reinterpret_cast<Fp2S*>(&r)[0] = p.x; reinterpret_cast<Fp2S*>(&r)[1] = p.y; reinterpret_cast<Fp2S*>(&r)[2] = fp2_zero();
    rec_store<6>(out, t, r);
}

// ------------------------------------------------------------------------------------------------
// kernels: scalar multiplication, sums
// ------------------------------------------------------------------------------------------------
template <class F> BLSMI_DEV Aff<F> load_aff(const u8* p);
template <> BLSMI_DEV G1Aff load_aff<FpS>(const u8* p) { return load_g1(p); }
template <> BLSMI_DEV G2Aff load_aff<Fp2S>(const u8* p) { return load_g2(p); }
BLSMI_DEV void store_aff(u8* p, const G1Aff& a) { store_g1(p, a); }
BLSMI_DEV void store_aff(u8* p, const G2Aff& a) { store_g2(p, a); }

// Fixed 4-bit-window scalar multiplication (BASELINE config 3).  The reference multiplies bit-serially
// (g1.go:80-90, g2.go:92-102: 255 doublings + one addition per set bit); here each lane builds the table
// {0, P, 2P, ..., 15P} in its scratch (per-lane indexed), then per nibble does four doublings and ONE addition:
// 252 doublings + 63 + 14 additions, uniform control flow for all 64 lanes.  Same group element, so the
// affine output is identical to the reference's.
template <class F, int PB>
__device__ void mul_batch_body(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    const Aff<F> p = load_aff<F>(pts + pt_stride * tt);                  // stride 0: one common base point (PrivToPub)
    const u32* s32 = reinterpret_cast<const u32*>(scalars + 32 * tt);
    Jac<F> tab[16];
    tab[0] = jac_zero<F>();
    tab[1] = to_jac(p);
    for (int j = 2; j < 16; j++) tab[j] = jac_add_affine(tab[j - 1], p);
    Jac<F> res = jac_zero<F>();
    for (int w = 0; w < 8; w++) {                                        // big-endian scalar: word 0 is most significant
        const u32 kw = __builtin_bswap32(s32[w]);
        for (int nib = 7; nib >= 0; nib--) {
            if (w | (7 - nib)) { res = jac_double(res); res = jac_double(res); res = jac_double(res); res = jac_double(res); }
            res = jac_add(res, tab[(kw >> (4 * nib)) & 15]);
        }
    }
    const Aff<F> a = jac_to_affine(res);
    if (t < n) { store_aff(out + (size_t)PB * t, a); out_inf[t] = a.inf ? 1 : 0; }
}
KERNEL2 k_g1_mul(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_batch_body<FpS, 96>(pts, pt_stride, scalars, out, out_inf, n); }
KERNEL k_g2_mul(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_batch_body<Fp2S, 192>(pts, pt_stride, scalars, out, out_inf, n); }

// Point sums: level 0 reads affine bytes pairwise into Jacobian SoA; later levels halve the array.
template <class F> struct jac_words { static constexpr int value = sizeof(F) / sizeof(FpS) * 3; };
template <class F>
BLSMI_DEV void jac_soa_store(i32* buf, size_t n, size_t t, const Jac<F>& p) {
    constexpr int W = jac_words<F>::value;
    const FpS* c = reinterpret_cast<const FpS*>(&p);
#pragma unroll
    for (int e = 0; e < W; e++) soa_store(buf, n, t, e, c[e]);
    buf[(size_t)W * NL * n + t] = p.inf;
}
template <class F>
BLSMI_DEV Jac<F> jac_soa_load(const i32* buf, size_t n, size_t t) {
    constexpr int W = jac_words<F>::value;
    Jac<F> p;
    FpS* c = reinterpret_cast<FpS*>(&p);
#pragma unroll
    for (int e = 0; e < W; e++) c[e] = soa_load(buf, n, t, e);
    p.inf = buf[(size_t)W * NL * n + t];
    return p;
}
template <class F, int PB>
__device__ void sum_level0_body(const u8* pts, const u8* in_inf, i32* buf, size_t n, size_t half) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= half) return;
    Aff<F> a = load_aff<F>(pts + (size_t)PB * t);
    if (in_inf && in_inf[t]) a.inf = -1;
    Jac<F> r = to_jac(a);
    if (t + half < n) {
        Aff<F> b = load_aff<F>(pts + (size_t)PB * (t + half));
        if (in_inf && in_inf[t + half]) b.inf = -1;
        r = jac_add_affine(r, b);
    }
    jac_soa_store(buf, half, t, r);
}
KERNEL2 k_g1_sum0(const u8* pts, const u8* in_inf, i32* buf, size_t n, size_t half) { sum_level0_body<FpS, 96>(pts, in_inf, buf, n, half); }
KERNEL k_g2_sum0(const u8* pts, const u8* in_inf, i32* buf, size_t n, size_t half) { sum_level0_body<Fp2S, 192>(pts, in_inf, buf, n, half); }
template <class F>
__device__ void sum_level_body(const i32* src, i32* dst, size_t n, size_t half) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= half) return;
    Jac<F> r = jac_soa_load<F>(src, n, t);
    if (t + half < n) r = jac_add(r, jac_soa_load<F>(src, n, t + half));
    jac_soa_store(dst, half, t, r);
}
KERNEL2 k_g1_sum(const i32* src, i32* dst, size_t n, size_t half) { sum_level_body<FpS>(src, dst, n, half); }
KERNEL k_g2_sum(const i32* src, i32* dst, size_t n, size_t half) { sum_level_body<Fp2S>(src, dst, n, half); }
template <class F, int PB>
__device__ void sum_final_body(const i32* src, u8* out, i32* out_inf) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Aff<F> a = jac_to_affine(jac_soa_load<F>(src, 1, 0));
    store_aff(out, a);
    *out_inf = a.inf ? 1 : 0;
}
KERNEL2 k_g1_sum_final(const i32* src, u8* out, i32* out_inf) { sum_final_body<FpS, 96>(src, out, out_inf); }
KERNEL k_g2_sum_final(const i32* src, u8* out, i32* out_inf) { sum_final_body<Fp2S, 192>(src, out, out_inf); }

#include "msm.inc"
#include "verify_kernels.inc"
#include "pair_kernels.inc"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {
// Host state. Every entry point leases one call context (a non-blocking stream, a grow-only scratch buffer and
// timing events) from a small pool, so calls from several OS threads (cgo pins one per goroutine call) run
// concurrently on separate HIP streams; temporaries come from the device's stream-ordered memory pool, whose
// release threshold is raised so freed blocks are reused by later calls instead of returning to the driver.
std::mutex g_mu;
std::condition_variable g_cv;
bool g_pair_layout = true;              // lane-pair pairing kernels (two lanes per tuple); BLSMI_LAYOUT=single for one tuple per lane
bool g_ready = false;
int g_device = 0;
char g_version[160] = "blsmi 0.1 (uninitialised)";

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "blsmi: %s failed: %s\n", #x, hipGetErrorString(e_)); return BLSMI_E_HIP; } } while (0)

// Grow-only scratch for the Miller-loop -> final-exponentiation hand-off of one call context.
struct Workspace {
    void* p = nullptr; size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
struct Ctx {
    hipStream_t stream = nullptr;
    Workspace ws;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    bool busy = false;
};
constexpr int MAX_CTX = 16;
Ctx g_ctx[MAX_CTX];
int g_nctx = 4;                         // BLSMI_STREAMS
thread_local Ctx* tl_ctx = nullptr;
#define g_stream (tl_ctx->stream)
#define g_ws (tl_ctx->ws)
// G1/G2 generators in wire form, written once at init (read-only afterwards)
struct Gens { u8* g1 = nullptr; u8* g2 = nullptr; i32* lines = nullptr; } g_gens;
bool g_use_gen_lines = true;           // BLSMI_GEN_LINES=0 recomputes the generator's lines per tuple (A/B switch)
// optional per-kernel timing (HIP events on the launch stream) for bench.py's roofline object
bool g_profile = false;
float g_last_ms[2] = {0.f, 0.f};

int ensure_init(int device) {           // caller holds g_mu
    if (g_ready) return BLSMI_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return BLSMI_E_NODEVICE;
    if (device < 0 || device >= count) return BLSMI_E_ARG;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    snprintf(g_version, sizeof g_version, "blsmi 0.1 %s CUs=%d", prop.gcnArchName, prop.multiProcessorCount);
    g_device = device;
    const char* lay = getenv("BLSMI_LAYOUT");
    g_pair_layout = !(lay && std::string(lay) == "single");      // default: lane-pair kernels; BLSMI_LAYOUT=single selects one tuple per lane
    const char* ns = getenv("BLSMI_STREAMS");
    if (ns) { int v = atoi(ns); g_nctx = v < 1 ? 1 : (v > MAX_CTX ? MAX_CTX : v); }
    hipMemPool_t pool;
    HIPCHK(hipDeviceGetDefaultMemPool(&pool, device));
    uint64_t keep = UINT64_MAX;
    HIPCHK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    if (!g_gens.g1) {
        HIPCHK(hipMalloc((void**)&g_gens.g1, 96)); HIPCHK(hipMalloc((void**)&g_gens.g2, 192));
        hipLaunchKernelGGL(k_write_generators, dim3(1), dim3(WG), 0, nullptr, g_gens.g1, g_gens.g2);
        HIPCHK(hipMalloc((void**)&g_gens.lines, sizeof(i32) * 68 * 3 * 2 * NL));
        hipLaunchKernelGGL(k_prepare_generator_lines, dim3(1), dim3(WG), 0, nullptr, (const u8*)g_gens.g2, g_gens.lines);
        HIPCHK(hipGetLastError());
        HIPCHK(hipDeviceSynchronize());
    }
    const char* gl = getenv("BLSMI_GEN_LINES");
    g_use_gen_lines = !(gl && std::string(gl) == "0");
    g_ready = true;
    return BLSMI_OK;
}
inline unsigned nblocks(size_t n) { return (unsigned)((n + WG - 1) / WG); }

// Lease of one call context for the duration of an entry point.
struct CtxLease {
    int rc = BLSMI_OK;
    CtxLease() {
        std::unique_lock<std::mutex> lk(g_mu);
        rc = ensure_init(g_device);
        if (rc) return;
        if (hipSetDevice(g_device) != hipSuccess) { rc = BLSMI_E_HIP; return; }     // the current device is per-thread state
        Ctx* c = nullptr;
        for (;;) {
            for (int i = 0; i < g_nctx && !c; i++) if (!g_ctx[i].busy) c = &g_ctx[i];
            if (c) break;
            g_cv.wait(lk);
        }
        if (!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = BLSMI_E_HIP; return; }
        c->busy = true;
        tl_ctx = c;
    }
    ~CtxLease() {
        if (!tl_ctx) return;
        { std::lock_guard<std::mutex> lk(g_mu); tl_ctx->busy = false; }
        tl_ctx = nullptr;
        g_cv.notify_one();
    }
};

// A caller-supplied stream (the *_dev entry points) stands in for the leased context's stream for one call.
struct UseStream {
    hipStream_t saved;
    explicit UseStream(void* s) : saved(tl_ctx->stream) { if (s) tl_ctx->stream = (hipStream_t)s; }
    ~UseStream() { tl_ctx->stream = saved; }
};

// RAII device temporary from the stream-ordered pool of the leased context's stream
struct DBuf {
    void* p = nullptr;
    hipStream_t s = nullptr;
    hipError_t alloc(size_t bytes) { s = g_stream; return hipMallocAsync(&p, bytes ? bytes : 1, s); }
    ~DBuf() { if (p) (void)hipFreeAsync(p, s); }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};
}  // namespace

#define LOCK_AND_INIT() CtxLease lease_; if (lease_.rc) return lease_.rc;

#define BLSMI_API extern "C" __attribute__((visibility("default")))

BLSMI_API int blsmi_init(int device) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ready && device != g_device) return BLSMI_E_ARG;
    return ensure_init(device);
}
BLSMI_API void blsmi_shutdown(void) {
    std::unique_lock<std::mutex> lk(g_mu);
    if (!g_ready) return;
    for (;;) {                          // wait for calls in flight
        bool busy = false;
        for (int i = 0; i < MAX_CTX; i++) busy |= g_ctx[i].busy;
        if (!busy) break;
        g_cv.wait(lk);
    }
    (void)hipSetDevice(g_device);
    for (int i = 0; i < MAX_CTX; i++) {
        Ctx& c = g_ctx[i];
        if (!c.stream) continue;
        (void)hipStreamSynchronize(c.stream);
        c.ws.release();
        for (auto& e : c.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        (void)hipStreamDestroy(c.stream);
        c.stream = nullptr;
    }
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, g_device) == hipSuccess) (void)hipMemPoolTrimTo(pool, 0);
    g_ready = false;
}
BLSMI_API const char* blsmi_version(void) { return g_version; }

// ---- pairing ------------------------------------------------------------------------------------
static int pairing_dev(const void* d_g1, const void* d_g2, void* d_out, size_t n, hipStream_t s, int mode) {
    if (n == 0) return BLSMI_OK;
    HIPCHK(g_ws.reserve(sizeof(i32) * 12 * NL * n));
    i32* f = reinterpret_cast<i32*>(g_ws.p);
    const bool prof = g_profile;
    if (prof && !tl_ctx->ev[0]) for (auto& e : tl_ctx->ev) HIPCHK(hipEventCreate(&e));
    if (prof) HIPCHK(hipEventRecord(tl_ctx->ev[0], s));
    const unsigned pblocks = (unsigned)((n + PT - 1) / PT);
    if (g_pair_layout) hipLaunchKernelGGL(k_miller1_pair, dim3(pblocks), dim3(WG), 0, s, (const u8*)d_g1, (const u8*)d_g2, f, n);
    else hipLaunchKernelGGL(k_miller1, dim3(nblocks(n)), dim3(WG), 0, s, (const u8*)d_g1, (const u8*)d_g2, f, n);
    if (prof) HIPCHK(hipEventRecord(tl_ctx->ev[1], s));
    if (g_pair_layout) hipLaunchKernelGGL(k_final_exp_pair, dim3(pblocks), dim3(WG), 0, s, (const i32*)f, (u64*)d_out, n, mode);
    else hipLaunchKernelGGL(k_final_exp, dim3(nblocks(n)), dim3(WG), 0, s, (const i32*)f, (u64*)d_out, n, mode);
    if (prof) HIPCHK(hipEventRecord(tl_ctx->ev[2], s));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));      // blocking entry point: results are ready on return
    if (prof) {
        HIPCHK(hipEventElapsedTime(&g_last_ms[0], tl_ctx->ev[0], tl_ctx->ev[1]));
        HIPCHK(hipEventElapsedTime(&g_last_ms[1], tl_ctx->ev[1], tl_ctx->ev[2]));
    }
    return BLSMI_OK;
}
// Enable/disable per-kernel HIP-event timing of blsmi_pairing_batch[_dev]; read back with blsmi_last_kernel_ms.
BLSMI_API int blsmi_set_profiling(int on) {
    LOCK_AND_INIT();
    g_profile = on != 0;
    return BLSMI_OK;
}
BLSMI_API int blsmi_last_kernel_ms(float* miller_ms, float* final_exp_ms) {
    if (!miller_ms || !final_exp_ms) return BLSMI_E_ARG;
    *miller_ms = g_last_ms[0]; *final_exp_ms = g_last_ms[1];
    return BLSMI_OK;
}
BLSMI_API int blsmi_pairing_batch_dev(const void* d_g1, const void* d_g2, void* d_out, size_t n, void* stream) {
    if (n && (!d_g1 || !d_g2 || !d_out)) return BLSMI_E_ARG;
    LOCK_AND_INIT();
    UseStream us(stream);
    return pairing_dev(d_g1, d_g2, d_out, n, g_stream, 0);
}
static int pairing_host(const uint8_t* g1, const uint8_t* g2, uint64_t* out, size_t n, int mode) {
    if (n && (!g1 || !g2 || !out)) return BLSMI_E_ARG;
    LOCK_AND_INIT();
    if (n == 0) return BLSMI_OK;
    DBuf a, b, o;
    HIPCHK(a.alloc(96 * n)); HIPCHK(b.alloc(192 * n)); HIPCHK(o.alloc(576 * n));
    HIPCHK(hipMemcpyAsync(a.p, g1, 96 * n, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(b.p, g2, 192 * n, hipMemcpyHostToDevice, g_stream));
    int rc = pairing_dev(a.p, b.p, o.p, n, g_stream, mode);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(out, o.p, 576 * n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return BLSMI_OK;
}
BLSMI_API int blsmi_pairing_batch(const uint8_t* g1, const uint8_t* g2, uint64_t* out, size_t n) { return pairing_host(g1, g2, out, n, 0); }
BLSMI_API int blsmi_miller_loop_batch(const uint8_t* g1, const uint8_t* g2, uint64_t* out, size_t n) { return pairing_host(g1, g2, out, n, 1); }
BLSMI_API int blsmi_final_exponentiation_batch(const uint64_t* in, uint64_t* out, size_t n) {
    if (n && (!in || !out)) return BLSMI_E_ARG;
    LOCK_AND_INIT();
    if (n == 0) return BLSMI_OK;
    DBuf i, f, o;
    HIPCHK(i.alloc(576 * n)); HIPCHK(o.alloc(576 * n)); HIPCHK(f.alloc(sizeof(i32) * 12 * NL * n));
    HIPCHK(hipMemcpyAsync(i.p, in, 576 * n, hipMemcpyHostToDevice, g_stream));
    hipLaunchKernelGGL(k_fq12_from_m384, dim3(nblocks(n)), dim3(WG), 0, g_stream, i.as<u64>(), f.as<i32>(), n);
    hipLaunchKernelGGL(k_final_exp, dim3(nblocks(n)), dim3(WG), 0, g_stream, f.as<i32>(), o.as<u64>(), n, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, o.p, 576 * n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return BLSMI_OK;
}

// ---- unit-level ops -------------------------------------------------------------------------------
BLSMI_API int blsmi_debug_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* flag, size_t n) {
    int width = op < 16 ? 1 : op < 32 ? 2 : op < 48 ? 6 : op < 64 ? 12 : (op == BLSMI_OP_G1_DOUBLE || op == BLSMI_OP_G1_ADD || op == BLSMI_OP_SWU_G1) ? 3 : 6;
    if (n && (!a || !out)) return BLSMI_E_ARG;
    LOCK_AND_INIT();
    if (n == 0) return BLSMI_OK;
    const size_t bytes = (size_t)48 * width * n;
    DBuf da, db, dout, dflag;
    HIPCHK(da.alloc(bytes)); HIPCHK(db.alloc(bytes)); HIPCHK(dout.alloc(bytes)); HIPCHK(dflag.alloc(n));
    HIPCHK(hipMemcpyAsync(da.p, a, bytes, hipMemcpyHostToDevice, g_stream));
    if (b) HIPCHK(hipMemcpyAsync(db.p, b, bytes, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemsetAsync(dflag.p, 1, n, g_stream));
    dim3 g(nblocks(n)), w(WG);
    if (op < 16) hipLaunchKernelGGL(k_debug_fq, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), dflag.as<u8>(), n);
    else if (op < 32) hipLaunchKernelGGL(k_debug_fq2, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), dflag.as<u8>(), n);
    else if (op < 48) hipLaunchKernelGGL(k_debug_fq6, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), n);
    else if (op < 64) hipLaunchKernelGGL(k_debug_fq12, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), n);
    else if (op == BLSMI_OP_SWU_G1) hipLaunchKernelGGL(k_debug_swu_g1, g, w, 0, g_stream, da.as<u64>(), dout.as<u64>(), n);
    else if (op == BLSMI_OP_SWU_G2) hipLaunchKernelGGL(k_debug_swu_g2, g, w, 0, g_stream, da.as<u64>(), dout.as<u64>(), n);
    else hipLaunchKernelGGL(k_debug_curve, g, w, 0, g_stream, op, da.as<u64>(), db.as<u64>(), dout.as<u64>(), n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, g_stream));
    if (flag) HIPCHK(hipMemcpyAsync(flag, dflag.p, n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return BLSMI_OK;
}

// ---- scalar multiplication / sums ----------------------------------------------------------------
// pts == nullptr: every scalar multiplies the group generator (PrivToPub, g2pubs/bls.go:138-140, g1pubs/bls.go:144-146)
template <int PB, class K>
static int mul_batch(K kernel, const uint8_t* pts, const u8* d_gen, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n) {
    if (n && ((!pts && !d_gen) || !scalars || !out || !out_inf)) return BLSMI_E_ARG;
    LOCK_AND_INIT();
    if (n == 0) return BLSMI_OK;
    DBuf dp, ds, dout, dinf;
    HIPCHK(ds.alloc(32 * n)); HIPCHK(dout.alloc((size_t)PB * n)); HIPCHK(dinf.alloc(n));
    if (pts) { HIPCHK(dp.alloc((size_t)PB * n)); HIPCHK(hipMemcpyAsync(dp.p, pts, (size_t)PB * n, hipMemcpyHostToDevice, g_stream)); }
    HIPCHK(hipMemcpyAsync(ds.p, scalars, 32 * n, hipMemcpyHostToDevice, g_stream));
    hipLaunchKernelGGL(kernel, dim3(nblocks(n)), dim3(WG), 0, g_stream, pts ? dp.as<u8>() : d_gen, (size_t)(pts ? PB : 0), ds.as<u8>(), dout.as<u8>(), dinf.as<u8>(), n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)PB * n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemcpyAsync(out_inf, dinf.p, n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return BLSMI_OK;
}
BLSMI_API int blsmi_g1_mul_batch(const uint8_t* pts, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n) { if (n && !pts) return BLSMI_E_ARG; return mul_batch<96>(k_g1_mul, pts, nullptr, scalars, out, out_inf, n); }
BLSMI_API int blsmi_g2_mul_batch(const uint8_t* pts, const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n) { if (n && !pts) return BLSMI_E_ARG; return mul_batch<192>(k_g2_mul, pts, nullptr, scalars, out, out_inf, n); }
BLSMI_API int blsmi_g1_mul_generator_batch(const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n) { return mul_batch<96>(k_g1_mul, nullptr, g_gens.g1, scalars, out, out_inf, n); }
BLSMI_API int blsmi_g2_mul_generator_batch(const uint8_t* scalars, uint8_t* out, uint8_t* out_inf, size_t n) { return mul_batch<192>(k_g2_mul, nullptr, g_gens.g2, scalars, out, out_inf, n); }

// tree reduction of n affine points already on the device; result (affine bytes + inf flag) on the device
template <int PB, int W, class K0, class K1, class K2>
static int sum_dev(K0 k0, K1 k1, K2 kfinal, const u8* d_pts, const u8* d_inf, size_t n, u8* d_out, i32* d_out_inf, hipStream_t s) {
    const size_t words = (size_t)W * NL + 1;
    size_t half = (n + 1) / 2;
    DBuf b0, b1;
    HIPCHK(b0.alloc(sizeof(i32) * words * half)); HIPCHK(b1.alloc(sizeof(i32) * words * ((half + 1) / 2)));
    hipLaunchKernelGGL(k0, dim3(nblocks(half)), dim3(WG), 0, s, d_pts, d_inf, b0.as<i32>(), n, half);
    i32* src = b0.as<i32>(); i32* dst = b1.as<i32>();
    size_t cur = half;
    while (cur > 1) {
        const size_t h = (cur + 1) / 2;
        hipLaunchKernelGGL(k1, dim3(nblocks(h)), dim3(WG), 0, s, (const i32*)src, dst, cur, h);
        std::swap(src, dst);
        cur = h;
    }
    hipLaunchKernelGGL(kfinal, dim3(1), dim3(WG), 0, s, (const i32*)src, d_out, d_out_inf);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return BLSMI_OK;
}
template <int PB, int W, class K0, class K1, class K2>
static int sum_host(K0 k0, K1 k1, K2 kfinal, const uint8_t* pts, const uint8_t* in_inf, size_t n, uint8_t* out, int* out_inf) {
    if (!out || !out_inf || (n && !pts)) return BLSMI_E_ARG;
    if (n == 0) { memset(out, 0, PB); *out_inf = 1; return BLSMI_OK; }      // empty sum = infinity (g2pubs/bls.go:166, 181)
    LOCK_AND_INIT();
    DBuf dp, di, dout, dflag;
    HIPCHK(dp.alloc((size_t)PB * n)); HIPCHK(di.alloc(n)); HIPCHK(dout.alloc(PB)); HIPCHK(dflag.alloc(sizeof(i32)));
    HIPCHK(hipMemcpyAsync(dp.p, pts, (size_t)PB * n, hipMemcpyHostToDevice, g_stream));
    if (in_inf) HIPCHK(hipMemcpyAsync(di.p, in_inf, n, hipMemcpyHostToDevice, g_stream));
    int rc = sum_dev<PB, W>(k0, k1, kfinal, dp.as<u8>(), in_inf ? di.as<u8>() : nullptr, n, dout.as<u8>(), dflag.as<i32>(), g_stream);
    if (rc) return rc;
    i32 flag = 0;
    HIPCHK(hipMemcpyAsync(out, dout.p, PB, hipMemcpyDeviceToHost, g_stream)); HIPCHK(hipStreamSynchronize(g_stream));
    HIPCHK(hipMemcpyAsync(&flag, dflag.p, sizeof flag, hipMemcpyDeviceToHost, g_stream)); HIPCHK(hipStreamSynchronize(g_stream));
    *out_inf = flag;
    return BLSMI_OK;
}
BLSMI_API int blsmi_g1_sum(const uint8_t* pts, const uint8_t* in_inf, size_t n, uint8_t out[96], int* out_inf) { return sum_host<96, 3>(k_g1_sum0, k_g1_sum, k_g1_sum_final, pts, in_inf, n, out, out_inf); }
BLSMI_API int blsmi_g2_sum(const uint8_t* pts, const uint8_t* in_inf, size_t n, uint8_t out[192], int* out_inf) { return sum_host<192, 6>(k_g2_sum0, k_g2_sum, k_g2_sum_final, pts, in_inf, n, out, out_inf); }

// multi-scalar multiplication sum_i k_i * P_i.  Small batches: per-point windowed multiples feed the tree sum on the
// device (one launch of latency, ~6 ms up to 64k points).  From BLSMI_MSM_BUCKET_MIN points (default 2^17, where the
// two cross over) on: the bucket method of msm.inc, whose fixed tail (chunk reduction + 240 serial doublings) is ~6 ms.
constexpr int BLSMI_E_SKEW = -1000;    // internal: bucket method declined (unbalanced digits)
struct MsmKernels {
    void (*bucket)(const u8*, const u32*, const u32*, const u32*, i32*, size_t, int, size_t);
    void (*chunk)(const i32*, i32*, int, int, size_t, size_t);
    void (*fold)(const i32*, i32*, size_t, size_t, int);
    void (*final)(const i32*, int, int, u8*, i32*);
};
template <int PB, int W>
static int msm_bucket_dev(const MsmKernels& k, const u8* d_pts, const u8* d_scalars, size_t n, u8* d_out, i32* d_flag, hipStream_t s) {
    // 16-bit windows: n / 2^16 points per bucket, and a 255-bit scalar still fills 15 bits of the top window (a window
    // size that leaves the top window a few bits wide would pile every point into a handful of buckets there)
    const int c = 16;
    const int K = 16;                                                      // buckets per chunk lane: 2^12 chunk lanes per window keep every SIMD busy
    const int nwin = (256 + c - 1) / c;
    const size_t B1 = (size_t)1 << c, nb = B1 * nwin, per_win = B1 / K, nct = per_win * nwin;
    const size_t jw = (size_t)W * NL + 1;                                  // words of one Jacobian SoA record
    DBuf hist, offs, cursor, idx, buckets, ch0, ch1, dmax;
    HIPCHK(hist.alloc(sizeof(u32) * nb)); HIPCHK(offs.alloc(sizeof(u32) * nb)); HIPCHK(cursor.alloc(sizeof(u32) * nb)); HIPCHK(dmax.alloc(sizeof(u32)));
    HIPCHK(idx.alloc(sizeof(u32) * n * nwin)); HIPCHK(buckets.alloc(sizeof(i32) * jw * nb));
    HIPCHK(ch0.alloc(sizeof(i32) * jw * nct)); HIPCHK(ch1.alloc(sizeof(i32) * jw * ((per_win + 1) / 2) * nwin));
    HIPCHK(hipMemsetAsync(hist.p, 0, sizeof(u32) * nb, s));
    HIPCHK(hipMemsetAsync(dmax.p, 0, sizeof(u32), s));
    hipLaunchKernelGGL(k_msm_hist, dim3(nblocks(n)), dim3(WG), 0, s, d_scalars, n, c, nwin, hist.as<u32>());
    // Skewed scalars (many equal digits) would leave one lane adding a whole bucket by itself: beyond 2048 points in
    // any bucket the caller falls back to the per-point multiples, whose cost does not depend on the scalars.
    hipLaunchKernelGGL(k_msm_max, dim3(nblocks(nb)), dim3(WG), 0, s, (const u32*)hist.as<u32>(), nb, dmax.as<u32>());
    u32 biggest = 0;
    HIPCHK(hipMemcpyAsync(&biggest, dmax.p, sizeof biggest, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (biggest > 2048) return BLSMI_E_SKEW;
    hipLaunchKernelGGL(k_msm_scan, dim3(nwin), dim3(256), 0, s, (const u32*)hist.as<u32>(), offs.as<u32>(), cursor.as<u32>(), c);
    hipLaunchKernelGGL(k_msm_scatter, dim3(nblocks(n)), dim3(WG), 0, s, d_scalars, n, c, nwin, cursor.as<u32>(), idx.as<u32>());
    hipLaunchKernelGGL(k.bucket, dim3(nblocks(nb)), dim3(WG), 0, s, d_pts, (const u32*)idx.as<u32>(), (const u32*)offs.as<u32>(), (const u32*)hist.as<u32>(), buckets.as<i32>(), n, c, nb);
    hipLaunchKernelGGL(k.chunk, dim3(nblocks(nct)), dim3(WG), 0, s, (const i32*)buckets.as<i32>(), ch0.as<i32>(), c, K, nb, nct);
    i32* src = ch0.as<i32>(); i32* dst = ch1.as<i32>();
    size_t seg = per_win;
    while (seg > 1) {
        const size_t half = (seg + 1) / 2;
        hipLaunchKernelGGL(k.fold, dim3(nblocks(half * nwin)), dim3(WG), 0, s, (const i32*)src, dst, seg, half, nwin);
        std::swap(src, dst);
        seg = half;
    }
    hipLaunchKernelGGL(k.final, dim3(1), dim3(WG), 0, s, (const i32*)src, nwin, c, d_out, d_flag);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));                                       // temporaries die with this scope
    return BLSMI_OK;
}
template <int PB, int W, class KM, class K0, class K1, class K2>
static int msm_host(const MsmKernels& mk, KM kmul, K0 k0, K1 k1, K2 kfinal, const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t* out, int* out_inf) {
    if (!out || !out_inf || (n && (!pts || !scalars))) return BLSMI_E_ARG;
    if (n == 0) { memset(out, 0, PB); *out_inf = 1; return BLSMI_OK; }
    LOCK_AND_INIT();
    static const size_t bucket_min = []{ const char* v = getenv("BLSMI_MSM_BUCKET_MIN"); return v ? (size_t)strtoull(v, nullptr, 10) : (size_t)1 << 17; }();
    DBuf dp, ds, dout, dflag;
    HIPCHK(dp.alloc((size_t)PB * n)); HIPCHK(ds.alloc(32 * n)); HIPCHK(dout.alloc(PB)); HIPCHK(dflag.alloc(sizeof(i32)));
    HIPCHK(hipMemcpyAsync(dp.p, pts, (size_t)PB * n, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(ds.p, scalars, 32 * n, hipMemcpyHostToDevice, g_stream));
    int rc;
    rc = BLSMI_E_SKEW;
    if (n >= bucket_min) rc = msm_bucket_dev<PB, W>(mk, dp.as<u8>(), ds.as<u8>(), n, dout.as<u8>(), dflag.as<i32>(), g_stream);
    if (rc == BLSMI_E_SKEW) {
        DBuf dm, dinf;
        HIPCHK(dm.alloc((size_t)PB * n)); HIPCHK(dinf.alloc(n));
        hipLaunchKernelGGL(kmul, dim3(nblocks(n)), dim3(WG), 0, g_stream, dp.as<u8>(), (size_t)PB, ds.as<u8>(), dm.as<u8>(), dinf.as<u8>(), n);
        rc = sum_dev<PB, W>(k0, k1, kfinal, dm.as<u8>(), dinf.as<u8>(), n, dout.as<u8>(), dflag.as<i32>(), g_stream);
    }
    if (rc) return rc;
    i32 flag = 0;
    HIPCHK(hipMemcpyAsync(out, dout.p, PB, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemcpyAsync(&flag, dflag.p, sizeof flag, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    *out_inf = flag;
    return BLSMI_OK;
}
BLSMI_API int blsmi_g1_msm(const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t out[96], int* out_inf) {
    static const MsmKernels mk{k_g1_msm_bucket, k_g1_msm_chunk, k_g1_msm_fold, k_g1_msm_final};
    return msm_host<96, 3>(mk, k_g1_mul, k_g1_sum0, k_g1_sum, k_g1_sum_final, pts, scalars, n, out, out_inf);
}
BLSMI_API int blsmi_g2_msm(const uint8_t* pts, const uint8_t* scalars, size_t n, uint8_t out[192], int* out_inf) {
    static const MsmKernels mk{k_g2_msm_bucket, k_g2_msm_chunk, k_g2_msm_fold, k_g2_msm_final};
    return msm_host<192, 6>(mk, k_g2_mul, k_g2_sum0, k_g2_sum, k_g2_sum_final, pts, scalars, n, out, out_inf);
}

#include "verify_host.inc"

