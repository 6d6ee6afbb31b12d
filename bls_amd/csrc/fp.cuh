// fp.cuh -- Fq arithmetic for BLS12-381 on gfx950 (CDNA4), one field element per lane.
//
// Replaces, on the device, the reference's L0-L2 layers: MultiplyFQRepr / MontReduce
// (stub_fallback.go:11-116, primitivefuncs_amd64.s), FQRepr (fqrepr.go) and FQ (fq.go:37-338).
//
// MI355X-first representation (NOT the reference's 6 x u64 saturated limbs):
//   * 15 signed limbs of 27 bits in int32 VGPRs, Montgomery form with R = 2^405.
//   * products are accumulated column-wise in one int64 accumulator with v_mad_i64_i32.  A column
//     holds 15 a_i*b_j and 15 m_i*q_j terms; with 27-bit limbs that is 15*(La*Lb+1)*2^54 < 2^63 for
//     La*Lb <= 33, so there is NO carry chain anywhere in a multiplication.  On gfx950 every
//     VALU->SGPR->VALU carry hop costs wait states (profiles/r01_ubench*.log): saturated 32-bit
//     limbs with v_addc chains are the slow design here, v_mad_u64/i64 issues at the plain VALU rate.
//   * add/sub/neg are 15 independent VALU ops; limbs may grow ("lazy") and are brought back by a
//     carry-free parallel normalisation.  How far a value may grow is tracked in the TYPE:
//     Fp<L,V> promises |limb| <= L*(2^27+64) and |value| <= V*q.  Every operation computes its
//     result bound at compile time and static_asserts the int64/int32 head-room, so a formula that
//     could overflow does not compile.  R/q = 2^24.3 leaves so much room that values never need
//     reducing inside the pairing; Montgomery products come back to |value| < 2q by themselves.
//   * results are made canonical ([0,q), the reference's invariant fq.go:41-45) only where they
//     leave the device or are compared; canonical values re-packed to 6 x u64 with R = 2^384 are
//     bit-identical to the reference's in-memory FQ.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blsmi {

typedef int32_t i32;
typedef int64_t i64;
typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

// Two builds of this header.  Default: 15 x 27-bit limbs, R = 2^405 -- every kernel family but one.  With -DBLSMI_LIMBS28 (the lane-pair and
// lane-quad pairing kernels): 14 x 28-bit limbs, R = 2^392: 196 instead of 225 multiply-adds per product -- the lane-pair Fq2 product
// 3 236 against 3 738 ns per call (tools/ubench_core28.hip, profiles/r04_ubench_core28.log) -- paid for with head-room: R / q is 2 560
// instead of 2^24, so a stored value may be no larger than 11 q and a VALUE REDUCTION (fp_reduce) replaces the limb normalisation at
// some results.  The bound V counts units of q / VU so that a Montgomery product (|value| < 1.41 q) is "3 halves" rather than "2 q":
// with whole-q bounds the Fq6 products' outputs would need reducing, with halves they do not.  Every formula still computes its bounds
// at compile time; where an operand is too large for a product the reduction is inserted by the type system (fp_fit).
#ifdef BLSMI_LIMBS28
constexpr int NL = 14;                 // limbs
constexpr int LB = 28;                 // bits per limb
constexpr int VU = 2;                  // V counts halves of q
constexpr int LMAX = 7;                // L*(2^28+64) < 2^31
constexpr int LPROD_MAX = 8;           // 14*(La*Lb+1)*2^56*(1+eps) < 2^63  <=>  La*Lb <= 8
constexpr int VPROD_MAX = 4096;        // in units^2: |a*b|/R <= 1024 q^2 / 2^392 = 0.4 q: products land in (-0.4q, 1.4q)
constexpr int VSTORE = 22;             // storage bound: 11 q -- two sums of two stored values still multiply in one fused pass (2 * 44 * 44 <= VPROD_MAX)
#else
constexpr int NL = 15;                 // limbs
constexpr int LB = 27;                 // bits per limb
constexpr int VU = 1;                  // V counts multiples of q
constexpr int LMAX = 15;               // L*(2^27+64) < 2^31
constexpr int LPROD_MAX = 33;          // 15*(La*Lb+1)*2^54*(1+eps) < 2^63  <=>  La*Lb <= 33
constexpr int VPROD_MAX = 1 << 23;     // |a*b|/R <= 2^23 q^2 / 2^405 < 0.41 q: products land in (-0.41q, 1.41q)
constexpr int VSTORE = 256;
#endif
constexpr i32 MASK = (1 << LB) - 1;
constexpr int VMAX = 1 << 12;          // top limb = value/2^(LB (NL-1)) stays far inside int32
constexpr int VMUL = VU + (VU + 1) / 2;    // bound of a Montgomery product: 2 q, or 3 halves of q (|value| < 1.41 q either way)
constexpr int VRED = 2 * VU + 1;           // bound of fp_reduce's output: (-1.01 q, 2.01 q)
#define BLSMI_DEV __device__ __forceinline__
// The hash kernels of a mid-size Verify are dependent chains on few waves, and the signature side's Miller kernel of the same call runs beside them on a side
// stream (verify_host.inc: verify_sig_side_start).  Two waves of a SIMD share its VALU issue by priority, then AGE (MI355X guide, "two waves per SIMD"): a hash
// wave that happens to be the younger one gets the leftover slots and its chain stretches 1.6x (k_hash_g2_front 1.12 -> 1.77 ms), an older one is nearly
// unimpeded -- which of the two happened was up to the dispatch order (g1pubs, 3 072 tuples: 4.33 or 4.97 ms a call).  The hash kernels therefore raise their
// priority once at entry; the Miller kernel beside them fills the slots the chains leave.  -DBLSMI_HASH_PRIO=0: off (A/B).
#ifndef BLSMI_HASH_PRIO
#define BLSMI_HASH_PRIO 2
#endif
__device__ __forceinline__ void hash_prio() {
#if BLSMI_HASH_PRIO
    __builtin_amdgcn_s_setprio(BLSMI_HASH_PRIO);
#endif
}

template <int L_, int V_>
struct Fp {
    static constexpr int L = L_, V = V_;
    i32 v[NL];
};
using FpS = Fp<1, VSTORE>;             // storage type: (near-)normalised limbs, |value| <= VSTORE q / VU
using FpC = Fp<1, VU>;                 // canonical: limbs in [0,2^LB), value in [0,q)

}  // namespace blsmi
#include "tower_fwd.cuh"
#ifdef BLSMI_LIMBS28
#include "consts28.cuh"
#else
#include "consts.cuh"
#endif
namespace blsmi {

// widen the promised bounds (never narrows)
template <int L2, int V2, int L, int V>
BLSMI_DEV Fp<L2, V2> fp_relabel(const Fp<L, V>& a) {
    static_assert(L2 >= L && V2 >= V, "relabel may only widen bounds");
    Fp<L2, V2> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = a.v[i];
    return r;
}

// Carry-free parallel normalisation: every limb keeps its low 27 bits and receives the (signed)
// overflow of the limb below.  Value unchanged; limbs 0..13 end in [-16, 2^27+16), top limb signed.
template <int L, int V>
BLSMI_DEV Fp<1, V> fp_norm(const Fp<L, V>& x) {
    static_assert(L <= LMAX, "limb bound exceeded before norm");
    Fp<1, V> r;
    if constexpr (L == 1) {
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = x.v[i];
    } else {
        r.v[0] = x.v[0] & MASK;
#pragma unroll
        for (int i = 1; i < NL - 1; i++) r.v[i] = (x.v[i] & MASK) + (x.v[i - 1] >> LB);
        r.v[NL - 1] = x.v[NL - 1] + (x.v[NL - 2] >> LB);
    }
    return r;
}
template <bool C, int L, int V>
BLSMI_DEV auto fp_norm_if(const Fp<L, V>& x) {
    if constexpr (C) return fp_norm(x); else return x;
}

// ---- add / sub / neg / small multiples: 15 independent VALU ops (fq.go:64-67, 82-87, 121-143) ----
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto fp_add(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) {
    if constexpr (La + Lb > LMAX) {
        if constexpr (La >= Lb) return fp_add(fp_norm(a), fp_norm_if<(1 + Lb > LMAX)>(b));
        else return fp_add(fp_norm_if<(La + 1 > LMAX)>(a), fp_norm(b));
    } else {
        static_assert(Va + Vb <= VMAX, "value bound exceeded: insert fp_reduce");
        Fp<La + Lb, Va + Vb> r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = a.v[i] + b.v[i];
        return r;
    }
}
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto fp_sub(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) {
    if constexpr (La + Lb > LMAX) {
        if constexpr (La >= Lb) return fp_sub(fp_norm(a), fp_norm_if<(1 + Lb > LMAX)>(b));
        else return fp_sub(fp_norm_if<(La + 1 > LMAX)>(a), fp_norm(b));
    } else {
        static_assert(Va + Vb <= VMAX, "value bound exceeded: insert fp_reduce");
        Fp<La + Lb, Va + Vb> r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = a.v[i] - b.v[i];
        return r;
    }
}
template <int L, int V>
BLSMI_DEV Fp<L, V> fp_neg(const Fp<L, V>& a) {
    Fp<L, V> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = -a.v[i];
    return r;
}
template <int K, int L, int V>
BLSMI_DEV auto fp_muls(const Fp<L, V>& a) {     // multiply by a small positive constant K
    if constexpr (K > LMAX) {                   // (28-bit limbs: 8 and 12 exceed the limb head-room even on a normalised value) K = 4 * (K / 4)
        static_assert(K % 4 == 0 && K / 4 <= LMAX, "factor the constant");
        return fp_muls<K / 4>(fp_norm(fp_muls<4>(a)));
    } else if constexpr (L * K > LMAX) return fp_muls<K>(fp_norm(a));
    else {
        static_assert(V * K <= VMAX, "value bound exceeded: insert fp_reduce");
        Fp<L * K, V * K> r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = a.v[i] * K;
        return r;
    }
}
template <int L, int V> BLSMI_DEV auto fp_dbl(const Fp<L, V>& a) { return fp_muls<2>(a); }

// ---- value reduction: subtract round(value/q)*q with exact carries -> value in (-1.01q, 2.01q) ------
// The quotient is estimated in fp32 from the two top limbs (value / 2^351); un-normalised lower limbs
// (|limb| <= 15 * 2^27) move the estimate by < 2^-25, and the exact carry pass below normalises anyway.
template <int L, int V>
BLSMI_DEV Fp<1, VRED> fp_reduce(const Fp<L, V>& x) {
    static_assert(V <= VMAX, "value bound exceeded");
    static_assert(L <= LMAX, "limb bound exceeded before reduce");
    const Fp<L, V>& y = x;
    const float top = (float)y.v[NL - 1] * (float)(1 << LB) + (float)y.v[NL - 2];
    const i32 k = (i32)floorf(top * BLSMI_Q_TOP2_INV);
    Fp<1, VRED> r;
    i64 c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        c += (i64)y.v[i] - (i64)k * C_Q[i];
        r.v[i] = (i32)c & MASK;
        c >>= LB;
    }
    c += (i64)y.v[NL - 1] - (i64)k * C_Q[NL - 1];
    r.v[NL - 1] = (i32)c;
    return r;
}
// fp_fit<VT>(x): x itself when its value bound is at most VT, else its reduction -- how products make their operands fit
template <int VT, int L, int V>
BLSMI_DEV auto fp_fit(const Fp<L, V>& x) {
    if constexpr (V <= VT) return x; else { static_assert(VRED <= VT, "no reduction can make this operand fit"); return fp_reduce(x); }
}
constexpr int isqrt_floor(int n) { int r = 0; while ((long long)(r + 1) * (r + 1) <= n) r++; return r; }
// ---- Montgomery multiplication (fq.go:70-79 = MultiplyFQRepr + MontReduce + reduceAssign) ----------
// Product-scanning: column k accumulates a_i*b_(k-i) and m_i*q_(k-i) in ONE int64 (v_mad_i64_i32),
// m_k = -acc/q mod 2^27 zeroes the low limb, the column is retired by an arithmetic shift.
// Output: limbs 0..13 in [0,2^27), signed top limb, value in (-0.41q, 1.41q) (|value| <= 2q).
// Kept out of line (vector-typed arguments travel in VGPRs v0-v29): the ~4 KB body stays hot in
// the instruction cache while tower code shrinks to call sequences.
typedef i32 vlimbs __attribute__((ext_vector_type(NL)));

BLSMI_DEV vlimbs fp_mul_body(vlimbs a, vlimbs b) {
    i32 m[NL];
    vlimbs r;
    i64 acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (i64)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)m[i] * C_Q[k - i];
        m[k] = (i32)((u32)(i32)acc * BLSMI_QINV) & MASK;
        acc += (i64)m[k] * C_Q[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (i64)a[i] * b[k - i];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (i64)m[i] * C_Q[k - i];
        r[k - NL] = (i32)acc & MASK;
        acc >>= LB;
    }
    r[NL - 1] = (i32)acc;
    return r;
}
// Squaring (fq.go:151-198): off-diagonal products once, against the doubled operand.
BLSMI_DEV vlimbs fp_sqr_body(vlimbs a) {
    i32 m[NL], a2[NL];
    vlimbs r;
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a[i] * 2;
    i64 acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (i64)a[i] * a2[k - i];
        if (k % 2 == 0) acc += (i64)a[k / 2] * a[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)m[i] * C_Q[k - i];
        m[k] = (i32)((u32)(i32)acc * BLSMI_QINV) & MASK;
        acc += (i64)m[k] * C_Q[0];
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; 2 * i < k; i++) acc += (i64)a[i] * a2[k - i];
        if (k % 2 == 0) acc += (i64)a[k / 2] * a[k / 2];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (i64)m[i] * C_Q[k - i];
        r[k - NL] = (i32)acc & MASK;
        acc >>= LB;
    }
    r[NL - 1] = (i32)acc;
    return r;
}
__device__ __noinline__ vlimbs fp_mul_core(vlimbs a, vlimbs b) { return fp_mul_body(a, b); }
__device__ __noinline__ vlimbs fp_sqr_core(vlimbs a) { return fp_sqr_body(a); }

template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto fp_mul(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) {
    if constexpr (La * Lb > LPROD_MAX) {
        if constexpr (La >= Lb) return fp_mul(fp_norm(a), b);
        else return fp_mul(a, fp_norm(b));
    } else if constexpr ((long long)Va * Vb > VPROD_MAX) {               // too large for one Montgomery pass: reduce the larger operand (28-bit limbs only)
        if constexpr (Va >= Vb) return fp_mul(fp_reduce(a), b); else return fp_mul(a, fp_reduce(b));
    } else {
        vlimbs x, y;
#pragma unroll
        for (int i = 0; i < NL; i++) { x[i] = a.v[i]; y[i] = b.v[i]; }
        vlimbs z = fp_mul_core(x, y);
        Fp<1, VMUL> r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = z[i];
        return r;
    }
}
template <int L, int V>
BLSMI_DEV auto fp_sqr(const Fp<L, V>& a) {
    if constexpr (2 * L * L > LPROD_MAX) return fp_sqr(fp_norm(a));
    else if constexpr ((long long)V * V > VPROD_MAX) return fp_sqr(fp_reduce(a));
    else {
        vlimbs x;
#pragma unroll
        for (int i = 0; i < NL; i++) x[i] = a.v[i];
        vlimbs z = fp_sqr_core(x);
        Fp<1, VMUL> r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.v[i] = z[i];
        return r;
    }
}

// bring any value into the storage type with the least work
template <int L, int V>
BLSMI_DEV FpS fp_store(const Fp<L, V>& x) {
    if constexpr (V > FpS::V) return fp_relabel<1, FpS::V>(fp_reduce(x));
    else return fp_relabel<1, FpS::V>(fp_norm(x));
}

// ---- canonical form [0,q) with limbs in [0,2^27) (the reference's invariant, fq.go:41-45) --------
// one exact pass adding (addq ? q : 0) - (subq ? q : 0); masks are all-ones / zero
BLSMI_DEV void limbs_addsub_q(const i32 in[NL], i32 out[NL], i32 addmask, i32 submask) {
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const i32 s = in[i] + (C_Q[i] & addmask) - (C_Q[i] & submask) + c;
        out[i] = (i < NL - 1) ? (s & MASK) : s;
        c = s >> LB;
    }
}
template <int L, int V>
BLSMI_DEV FpC fp_canon(const Fp<L, V>& x) {
    Fp<1, VRED> r = fp_reduce(x);                         // limbs 0..NL-2 in [0,2^LB), value in (-1.01q, 2.01q)
    i32 t[NL], d[NL];
    limbs_addsub_q(r.v, t, r.v[NL - 1] >> 31, 0);          // + q if negative  -> [0, 2.01q)
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {                // - q while >= q (at most twice)
        limbs_addsub_q(t, d, 0, -1);
        const i32 lt = d[NL - 1] >> 31;                    // all-ones iff t < q
#pragma unroll
        for (int i = 0; i < NL; i++) t[i] = (t[i] & lt) | (d[i] & ~lt);
    }
    FpC o;
#pragma unroll
    for (int i = 0; i < NL; i++) o.v[i] = t[i];
    return o;
}
template <int L, int V>
BLSMI_DEV bool fp_is_zero(const Fp<L, V>& x) {            // fq.go:146-148
    FpC c = fp_canon(x);
    i32 o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= c.v[i];
    return o == 0;
}
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV bool fp_eq(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) { return fp_is_zero(fp_sub(a, b)); }   // fq.go:116-118

// lane-wise select without v_cndmask: r = m ? a : b with m all-ones / zero
template <int L, int V>
BLSMI_DEV Fp<L, V> fp_select(i32 m, const Fp<L, V>& a, const Fp<L, V>& b) {
    Fp<L, V> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = (a.v[i] & m) | (b.v[i] & ~m);
    return r;
}
BLSMI_DEV FpS fp_zero() {
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = 0;
    return r;
}
BLSMI_DEV FpS fp_one() { return C_ONE; }

// ---- exponentiation by a fixed public exponent (fq.go:96-113), bits MSB-first from constant memory.
// The loop is rolled: two call sites, uniform control flow (the exponent is the same for all lanes).
// Fixed 4-bit windows, MSB first: nbits-ish squarings + one multiplication per non-zero window + 14 for the table
// {a^1..a^15} (per-lane scratch, indexed by the uniform window value): ~480 instead of ~610 field multiplications
// for the 380-bit exponents q-2, (q-3)/4 used by inversion and square roots.
BLSMI_DEV vlimbs fp_pow_core(vlimbs a, const u32* ebits, int nbits) {
    vlimbs tab[16];
    tab[1] = a;
    for (int j = 2; j < 16; j++) tab[j] = (j & 1) ? fp_mul_core(tab[j - 1], a) : fp_sqr_core(tab[j >> 1]);
    const int nw = (nbits + 3) >> 2;
    int i = nw - 1;
    vlimbs res = tab[(ebits[(4 * i) >> 5] >> ((4 * i) & 31)) & 15];       // top window: non-zero (the top bit is set)
    for (i = nw - 2; i >= 0; i--) {
        res = fp_sqr_core(fp_sqr_core(fp_sqr_core(fp_sqr_core(res))));
        const u32 w = (ebits[(4 * i) >> 5] >> ((4 * i) & 31)) & 15;
        if (w) res = fp_mul_core(res, tab[w]);
    }
    return res;
}
// a^((q-3)/4), the exponent of every square root (fq.go:203-217) and of the fused SWU maps: a sliding-window chain computed by
// gen_consts.py (5-bit windows over the odd powers a^1..a^31: 376 squarings + 81 products against 376 + 109 for the fixed windows above)
BLSMI_DEV vlimbs fp_pow_qm3o4_core(vlimbs a) {
    vlimbs tab[16];
    const vlimbs a2 = fp_sqr_core(a);
    tab[0] = a;
    for (int j = 1; j < 16; j++) tab[j] = fp_mul_core(tab[j - 1], a2);
    vlimbs res = tab[C_QM3O4_CHAIN[0] & 0xff];
    for (int k = 1; k < BLSMI_QM3O4_CHAIN_OPS; k++) {
        const u32 op = C_QM3O4_CHAIN[k];
        for (u32 sq = op >> 8; sq; sq--) res = fp_sqr_core(res);
        if ((op & 0xff) != 0xff) res = fp_mul_core(res, tab[op & 0xff]);
    }
    return res;
}
template <int L, int V>
BLSMI_DEV FpS fp_pow_qm3o4(const Fp<L, V>& a) {
    const FpS base = fp_store(a);
    vlimbs x;
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = base.v[i];
    const vlimbs z = fp_pow_qm3o4_core(x);
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = z[i];
    return r;
}
template <int L, int V>
BLSMI_DEV FpS fp_pow_const(const Fp<L, V>& a, const u32* ebits, int nbits) {
    const FpS base = fp_store(a);                          // L = 1, |value| <= 256 q: products stay valid
    vlimbs x;
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = base.v[i];
    const vlimbs z = fp_pow_core(x, ebits, nbits);
    FpS r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = z[i];
    return r;
}
// Inverse.  The reference runs a data-dependent binary extended Euclid (fq.go:224-266); a^(q-2) by the windowed power
// above is the obvious divergence-free substitute and costs ~480 multiplications (274 k instructions).  This is the
// Bernstein-Yang "safegcd" iteration instead (divsteps with 2x2 transition matrices, the formulation popularised by
// libsecp256k1's modinv), laid out for the 27-bit signed limbs: every batch derives a transition matrix from the low
// 27 bits of (f, g) with 27 branch-free divsteps, then applies it to (f, g) (exact division by 2^27) and to (d, e)
// (division mod q).  34 batches of 27 divsteps = 918 cover the bound of 878 divsteps for a 381-bit modulus
// (floor((45907 * 381 + 26313) / 19929), the half-delta variant started at delta = 1/2); control flow and
// instruction stream are the same for every lane (~21 k instructions).  inverse(0) = 0, as the power gave.
// Input: any representative; output: the Montgomery form of the inverse of the value the input stands for.
BLSMI_DEV FpS fp_inv_fermat(const FpS& a) { return fp_pow_const(a, C_QM2, BLSMI_QM2_BITS); }
__device__ __noinline__ vlimbs fp_inv_core(vlimbs a_canon) {
    i32 f[NL], g[NL], d[NL], e[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) { f[i] = C_Q[i]; g[i] = a_canon[i]; d[i] = 0; e[i] = 0; }
    e[0] = 1;
    i32 zeta = -1;                                                         // -(delta + 1/2), delta starts at 1/2
    for (int it = 0; it < (878 + LB - 1) / LB + 1; it++) {               // 34 batches of 27 divsteps (33 of 28): >= the 878-divstep bound
        // transition matrix of the next 27 divsteps from the low bits of f and g
        u32 u = 1, v = 0, qq = 0, r = 1;
        u32 fl = (u32)f[0] | ((u32)f[1] << LB), gl = (u32)g[0] | ((u32)g[1] << LB);
#pragma unroll
        for (int i = 0; i < LB; i++) {
            u32 c1 = (u32)(zeta >> 31);                                    // delta > 0
            const u32 c2 = 0u - (gl & 1u);                                 // g odd
            const u32 x = (fl ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // (f, u, v) negated when delta > 0
            gl += x & c2; qq += y & c2; r += z & c2;
            c1 &= c2;                                                      // swap case: delta > 0 and g odd
            zeta = (i32)((u32)zeta ^ c1) - 1;
            fl += gl & c1; u += qq & c1; v += r & c1;
            gl >>= 1; u <<= 1; v <<= 1;
        }
        const i32 mu = (i32)u, mv = (i32)v, mq = (i32)qq, mr = (i32)r;
        // (d, e) <- matrix * (d, e) / 2^27 mod q: a multiple of q is added so that the low 27 bits cancel
        {
            const i32 sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
            i32 md = (mu & sd) + (mv & se), me = (mq & sd) + (mr & se);
            i64 cd = (i64)mu * d[0] + (i64)mv * e[0], ce = (i64)mq * d[0] + (i64)mr * e[0];
            md -= (i32)((BLSMI_QINV_POS * (u32)cd + (u32)md) & (u32)MASK);
            me -= (i32)((BLSMI_QINV_POS * (u32)ce + (u32)me) & (u32)MASK);
            cd += (i64)C_Q[0] * md; ce += (i64)C_Q[0] * me;
            cd >>= LB; ce >>= LB;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                cd += (i64)mu * d[i] + (i64)mv * e[i] + (i64)C_Q[i] * md;
                ce += (i64)mq * d[i] + (i64)mr * e[i] + (i64)C_Q[i] * me;
                d[i - 1] = (i32)cd & MASK; cd >>= LB;
                e[i - 1] = (i32)ce & MASK; ce >>= LB;
            }
            d[NL - 1] = (i32)cd; e[NL - 1] = (i32)ce;
        }
        // (f, g) <- matrix * (f, g) / 2^27 (exact)
        {
            i64 cf = (i64)mu * f[0] + (i64)mv * g[0], cg = (i64)mq * f[0] + (i64)mr * g[0];
            cf >>= LB; cg >>= LB;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                cf += (i64)mu * f[i] + (i64)mv * g[i];
                cg += (i64)mq * f[i] + (i64)mr * g[i];
                f[i - 1] = (i32)cf & MASK; cf >>= LB;
                g[i - 1] = (i32)cg & MASK; cg >>= LB;
            }
            f[NL - 1] = (i32)cf; g[NL - 1] = (i32)cg;
        }
    }
    // g = 0 and f = +-1 now (f = q for a = 0, where d = 0): the inverse is sign(f) * d, somewhere in (-2q, 2q)
    const i32 sf = f[NL - 1] >> 31;
    vlimbs out;
#pragma unroll
    for (int i = 0; i < NL; i++) out[i] = (d[i] ^ sf) - sf;
    return out;
}
template <int L, int V>
BLSMI_DEV FpS fp_inv(const Fp<L, V>& a) {
    const FpC c = fp_canon(a);
    vlimbs x;
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = c.v[i];
    const vlimbs z = fp_inv_core(x);
    Fp<2, 3 * VU> r;                                                       // limb-wise negated limbs: |limb| < 2^LB + 1, |value| < 2q
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = z[i];
    // z = (a R)^-1 as a plain integer = a^-1 R^-1; one Montgomery product with R^3 gives a^-1 R
    return fp_store(fp_mul(fp_relabel<1, 3 * VU>(fp_norm(r)), C_R3));
}

// Square root (fq.go:203-217): a1 = a^((q-3)/4); a0 = a1^2 a; ok iff a0 != -1; root = a1*a.
template <int SPREAD, int L, int V> BLSMI_DEV FpS fp_pow_spread(const Fp<L, V>& x, const u32* ebits, int nbits);   // fp_row.cuh: one limb per lane; 1: wave-uniform x, 2: x uniform over each row of sixteen lanes
template <int WAVE = 0, int L, int V>
BLSMI_DEV FpS fp_sqrt(const Fp<L, V>& a, bool& ok) {
    const FpS as = fp_store(a);
    FpS a1;
    if constexpr (WAVE != 0) a1 = fp_pow_spread<WAVE>(as, C_QM3O4, BLSMI_QM3O4_BITS);
    else a1 = fp_pow_qm3o4(as);
    const auto a0 = fp_mul(fp_sqr(a1), as);
    ok = !fp_eq(a0, C_NEGONE);
    return fp_store(fp_mul(a1, as));
}

// ---- representation changes at the device boundary --------------------------------------------------
// 12 little-endian u32 words (a 384-bit integer) <-> 15 x 27-bit limbs
BLSMI_DEV void words_to_limbs(const u32 w[12], i32 l[NL]) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LB * i, j = bit >> 5, sh = bit & 31;
        u32 x = (j < 12) ? (w[j] >> sh) : 0u;
        if (sh > 32 - LB && j + 1 < 12) x |= w[j + 1] << (32 - sh);
        l[i] = (i32)(x & (u32)MASK);
    }
}
BLSMI_DEV void limbs_to_words(const i32 l[NL], u32 w[12]) {   // limbs must be canonical (non-negative, < 2^27)
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int bit = 32 * j, i = bit / LB, sh = bit % LB;
        u32 x = (u32)l[i] >> sh;
        if (i + 1 < NL) x |= (u32)l[i + 1] << (LB - sh);
        if (2 * LB - sh < 32 && i + 2 < NL) x |= (u32)l[i + 2] << (2 * LB - sh);
        w[j] = x;
    }
}
// normal-form integer (raw limbs, must be < q else it becomes 0 like FQReprToFQ fq.go:49-56) -> Montgomery
BLSMI_DEV FpS fp_from_words(const u32 w[12]) {
    Fp<1, VU> raw;
    words_to_limbs(w, raw.v);
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) c = (raw.v[i] - C_Q[i] + c) >> LB;
    const i32 valid = c;                                   // all-ones iff raw < q (raw - q borrows)
#pragma unroll
    for (int i = 0; i < NL; i++) raw.v[i] &= valid;
    return fp_relabel<1, FpS::V>(fp_mul(raw, C_R2));
}
// Montgomery -> canonical normal-form integer words
template <int L, int V>
BLSMI_DEV void fp_to_words(const Fp<L, V>& a, u32 w[12]) {
    FpC c = fp_canon(fp_mul(fp_store(a), C_RAW_ONE));
    limbs_to_words(c.v, w);
}
// Montgomery(2^405) -> the reference's in-memory FQ image: canonical x*2^384 mod q as 12 LE u32 (= 6 LE u64)
template <int L, int V>
BLSMI_DEV void fp_to_mont384_words(const Fp<L, V>& a, u32 w[12]) {
    FpC c = fp_canon(fp_mul(fp_store(a), C_TO_M384));
    limbs_to_words(c.v, w);
}
BLSMI_DEV FpS fp_from_mont384_words(const u32 w[12]) {
    Fp<1, 10 * VU> raw;                                    // a 384-bit word may exceed q (2^384 < 10 q)
    words_to_limbs(w, raw.v);
    return fp_relabel<1, FpS::V>(fp_mul(raw, C_FROM_M384));
}
// ---- the 27-bit-limb kernels' form of a value, at kernel boundaries ------------------------------------------------------------
// Buffers that cross between kernels (the Miller-loop -> final-exponentiation hand-off, the Fq12 product tree, prepared lines) hold
// field elements as the 15 x 27-bit limbs of the Montgomery(2^405) form -- the format of every kernel family but the 28-bit pairing
// kernels, which convert where they load and store (6 elements per lane and kernel: ~0.1 % of a pairing).  x R27 = x R28 2^13.
constexpr int NL_IO = 15;              // words per field element in those buffers (either build)
constexpr i32 MASK27 = (1 << 27) - 1;
#ifdef BLSMI_LIMBS28
template <int L2, int V2, int L, int V>
BLSMI_DEV Fp<L2, V2> fp_bound_of_constant(const Fp<L, V>& a) {            // a constant table entry is typed FpS but IS below q
    Fp<L2, V2> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = a.v[i];
    return r;
}
// 15 near-normalised signed limbs, |value| <= 256 q (what fp_store of the 27-bit build leaves) -> the same element here
BLSMI_DEV FpS fp_from_limbs27(const i32 l[NL_IO]) {
    i32 t[NL_IO];
    i32 c = 0;
#pragma unroll
    for (int i = 0; i < NL_IO - 1; i++) { const i32 s = l[i] + C_512Q_27[i] + c; t[i] = s & MASK27; c = s >> 27; }   // + 512 q: non-negative, exact carries
    t[NL_IO - 1] = l[NL_IO - 1] + C_512Q_27[NL_IO - 1] + c;
    Fp<1, 768 * VU> raw;                                                   // the integer a R27 + k q, below 2^392, re-sliced into 28-bit limbs
#pragma unroll
    for (int j = 0; j < NL; j++) {
        const int bit = LB * j, i = bit / 27, sh = bit % 27;
        u32 x = (u32)t[i] >> sh;
        if (i + 1 < NL_IO) x |= (u32)t[i + 1] << (27 - sh);
        raw.v[j] = (i32)(x & (u32)MASK);
    }
    return fp_relabel<1, FpS::V>(fp_mul(raw, fp_bound_of_constant<1, VU>(C_FROM27)));   // mont28(X, R28 / 2^13) = a R28
}
template <int L, int V>
BLSMI_DEV void fp_to_limbs27(const Fp<L, V>& a, i32 out[NL_IO]) {          // canonical a R27 mod q as 15 x 27-bit limbs
    const FpC c = fp_canon(fp_mul(fp_store(a), fp_bound_of_constant<1, VU>(C_TO27)));
#pragma unroll
    for (int i = 0; i < NL_IO; i++) {
        const int bit = 27 * i, j = bit / LB, sh = bit % LB;
        u32 x = (u32)c.v[j] >> sh;
        if (j + 1 < NL) x |= (u32)c.v[j + 1] << (LB - sh);
        out[i] = (i32)(x & (u32)MASK27);
    }
}
#else
BLSMI_DEV FpS fp_from_limbs27(const i32 l[NL_IO]) { FpS r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.v[i] = l[i];
    return r; }
template <int L, int V>
BLSMI_DEV void fp_to_limbs27(const Fp<L, V>& a, i32 out[NL_IO]) { const FpS s = fp_store(a);
#pragma unroll
    for (int i = 0; i < NL; i++) out[i] = s.v[i]; }
#endif
// big-endian 48-byte field element (g1.go:157-167 wire order) <-> words
BLSMI_DEV void be48_to_words(const u8* p, u32 w[12]) {
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const u8* b = p + 44 - 4 * j;
        w[j] = ((u32)b[0] << 24) | ((u32)b[1] << 16) | ((u32)b[2] << 8) | (u32)b[3];
    }
}
BLSMI_DEV void words_to_be48(const u32 w[12], u8* p) {
#pragma unroll
    for (int j = 0; j < 12; j++) {
        u8* b = p + 44 - 4 * j;
        b[0] = (u8)(w[j] >> 24); b[1] = (u8)(w[j] >> 16); b[2] = (u8)(w[j] >> 8); b[3] = (u8)w[j];
    }
}

}  // namespace blsmi
