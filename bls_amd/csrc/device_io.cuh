// device_io.cuh -- launch geometry and device-side tuple I/O shared by the kernel translation units
// (k_pairing_single.hip, k_pairing_pair.hip, k_hash.hip, k_curve.hip).  64-lane workgroups throughout.
// A one-tuple-per-lane pairing keeps f (180 words), R (90) and P, Q (90) per lane and wants the whole 512-entry
// register file, i.e. one wave per SIMD; G1-side kernels fit 256 registers and run two.
#pragma once
#include "../../include/blsmi.h"
#include "curve.cuh"

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
// Two forms.  *_io: ALWAYS the 15 x 27-bit limbs of the Montgomery(2^405) form, whatever this unit's own limbs are -- the format of
// buffers that cross between kernel families (the Miller-loop -> final-exponentiation hand-off, the Fq12 product tree, what the latency
// programs of k_lat.hip read): a 28-bit-limb unit converts here (fp.cuh: fp_to_limbs27 / fp_from_limbs27).  Plain soa_store / soa_load
// are the *_io forms unless the unit says BLSMI_SOA_NATIVE: then they keep the unit's own limbs (buffers no other unit reads: the
// MSM passes and point sums of k_curve.hip / k_msm_pair.hip).  With 27-bit limbs the two coincide.
BLSMI_DEV void soa_store_io(i32* buf, size_t n, size_t t, int e, const FpS& x) {
    i32 w[NL_IO];
    fp_to_limbs27(x, w);
#pragma unroll
    for (int j = 0; j < NL_IO; j++) buf[((size_t)e * NL_IO + j) * n + t] = w[j];
}
BLSMI_DEV FpS soa_load_io(const i32* buf, size_t n, size_t t, int e) {
    i32 w[NL_IO];
#pragma unroll
    for (int j = 0; j < NL_IO; j++) w[j] = buf[((size_t)e * NL_IO + j) * n + t];
    return fp_from_limbs27(w);
}
#ifdef BLSMI_SOA_NATIVE
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
#else
BLSMI_DEV void soa_store(i32* buf, size_t n, size_t t, int e, const FpS& x) { soa_store_io(buf, n, t, e, x); }
BLSMI_DEV FpS soa_load(const i32* buf, size_t n, size_t t, int e) { return soa_load_io(buf, n, t, e); }
#endif
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
// `any` (optional) accumulates the OR of the raw words: an all-zero affine record is the point at infinity
BLSMI_DEV FpS load_be48(const u8* p, u32* any = nullptr) {
    const u32* w32 = reinterpret_cast<const u32*>(p);
    u32 w[12];
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = __builtin_bswap32(w32[11 - j]);
    if (any) {
#pragma unroll
        for (int j = 0; j < 12; j++) *any |= w[j];
    }
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
// One FQ of the reference as it lies in the Go heap (fq.go:11-13: 6 LE u64, Montgomery 2^384) -- the coordinates of a
// bls.G1Projective / G2Projective handed over by the *_jac entry points.  The reference's arithmetic keeps every FQ below q
// (fq.go:37-45), so a limb image >= q cannot come out of it; one that arrives anyway is read the way FQReprToFQ reads an
// invalid repr (fq.go:49-56): as 0.  nz accumulates the OR of the (valid) words: nz == 0 <=> the element is 0 (FQ.IsZero);
// diff accumulates the OR of the differences to `expect` (12 words, or null for 0): diff == 0 <=> the element IS that value.
BLSMI_DEV FpS load_m384_checked(const u64* p, u32& nz, u32& diff, const u32* expect) {
    const u32* w32 = reinterpret_cast<const u32*>(p);
    u32 w[12];
    u32 borrow = 0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        w[j] = w32[j];
        const u64 d = (u64)w[j] - (u64)C_Q_WORDS[j] - borrow;
        borrow = (u32)(d >> 32) & 1u;
    }
    const u32 valid = 0u - borrow;                                         // all ones iff the integer is below q
#pragma unroll
    for (int j = 0; j < 12; j++) { w[j] &= valid; nz |= w[j]; diff |= w[j] ^ (expect ? expect[j] : 0u); }
    return fp_from_mont384_words(w);
}
// A whole in-memory point: x, y, z as NC = 1 (G1Projective, g1.go:252-256) or 2 (G2Projective, g2.go:298-302) FQ each.
// inf <=> z.IsZero() (g1.go:287-289, g2.go:325-327); *z_is_one (optional) <=> z is exactly FQOne / FQ2One, what ToProjective
// of a deserialised point leaves (g1.go:59-64) -- ToAffine then has nothing to invert.
template <class F>
BLSMI_DEV Jac<F> load_jac_m384(const u64* rec, bool* z_is_one = nullptr) {
    constexpr int NC = sizeof(F) / sizeof(FpS);
    Jac<F> j;
    FpS* c = reinterpret_cast<FpS*>(&j);
    u32 nz_xy = 0, nz_z = 0, d_xy = 0, d_z = 0;
#pragma unroll
    for (int e = 0; e < 2 * NC; e++) c[e] = load_m384_checked(rec + 6 * e, nz_xy, d_xy, nullptr);
#pragma unroll
    for (int e = 0; e < NC; e++) c[2 * NC + e] = load_m384_checked(rec + 6 * (2 * NC + e), nz_z, d_z, e == 0 ? C_ONE_M384_WORDS : nullptr);
    j.inf = nz_z ? 0 : -1;
    if (z_is_one) *z_is_one = d_z == 0;
    return j;
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

// Affine records: x || y big-endian.  The all-zero record is the library's encoding of the point at infinity (what
// store_g1 / store_g2 write for it; (0, 0) is on neither curve), so it is read back as infinity everywhere.
BLSMI_DEV G1Aff load_g1(const u8* p) { u32 any = 0; G1Aff a; a.x = load_be48(p, &any); a.y = load_be48(p + 48, &any); a.inf = any ? 0 : -1; return a; }
BLSMI_DEV G2Aff load_g2(const u8* p) {
    u32 any = 0;
    G2Aff a; a.x.c0 = load_be48(p, &any); a.x.c1 = load_be48(p + 48, &any); a.y.c0 = load_be48(p + 96, &any); a.y.c1 = load_be48(p + 144, &any); a.inf = any ? 0 : -1; return a;
}
BLSMI_DEV void store_g1(u8* p, const G1Aff& a) {
    if (a.inf) { u32* w = reinterpret_cast<u32*>(p); for (int i = 0; i < 24; i++) w[i] = 0; return; }
    store_be48(p, a.x); store_be48(p + 48, a.y);
}
BLSMI_DEV void store_g2(u8* p, const G2Aff& a) {
    if (a.inf) { u32* w = reinterpret_cast<u32*>(p); for (int i = 0; i < 48; i++) w[i] = 0; return; }
    store_be48(p, a.x.c0); store_be48(p + 48, a.x.c1); store_be48(p + 96, a.y.c0); store_be48(p + 144, a.y.c1);
}

// ... and back: an affine point as the in-memory Jacobian record with z = 1 (what G?Affine.ToProjective leaves, g1.go:59-64, g2.go:70-76);
// the point at infinity as (0, 1, 0), the reference's G?ProjectiveZero (g1.go:275, g2.go:313)
template <class F>
BLSMI_DEV void store_jac_m384(u64* rec, const Aff<F>& a) {
    constexpr int NC = sizeof(F) / sizeof(FpS);
    u32* w = reinterpret_cast<u32*>(rec);
    if (a.inf) {
        for (int i = 0; i < 36 * NC; i++) w[i] = 0;
        for (int j = 0; j < 12; j++) w[12 * NC + j] = C_ONE_M384_WORDS[j];
        return;
    }
    const FpS* c = reinterpret_cast<const FpS*>(&a);
#pragma unroll
    for (int e = 0; e < 2 * NC; e++) store_m384(rec + 6 * e, c[e]);
    for (int j = 0; j < 12 * NC; j++) w[24 * NC + j] = j < 12 ? C_ONE_M384_WORDS[j] : 0u;
}

// records of W Fq values in the Fq wire format (6 LE u64, Montgomery 2^384), used by the unit-level debug kernels
template <int W> struct Rec { FpS e[W]; };
template <int W> BLSMI_DEV Rec<W> rec_load(const u64* p, size_t t) { Rec<W> r; for (int i = 0; i < W; i++) r.e[i] = load_m384(p + (size_t)6 * (W * t + i)); return r; }
template <int W> BLSMI_DEV void rec_store(u64* p, size_t t, const Rec<W>& r) { for (int i = 0; i < W; i++) store_m384(p + (size_t)6 * (W * t + i), r.e[i]); }
template <class T, int W> BLSMI_DEV T& as(Rec<W>& r) { return *reinterpret_cast<T*>(&r); }

