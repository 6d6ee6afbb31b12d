// tower.cuh -- Fq2 / Fq6 / Fq12 on the device (replaces fq2.go:41-232, fq6.go:34-336, fq12.go:27-237).
// Same tower as the reference: Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(1+u)), Fq12 = Fq6[w]/(w^2-v).
// Every function returns the mathematically same element as the cited reference method; bounds
// (<L,V>, see fp.cuh) are propagated by the types, so `auto` results are "lazy" and are brought back
// to the storage types (Fp2S/Fp6S/Fp12S) with *_store() where a value is kept across a loop.
#pragma once
#include "fp.cuh"

namespace blsmi {

constexpr int imax(int a, int b) { return a > b ? a : b; }

template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto make_fp2(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) {
    constexpr int L = imax(La, Lb), V = imax(Va, Vb);
    Fp2<L, V> r;
    r.c0 = fp_relabel<L, V>(a);
    r.c1 = fp_relabel<L, V>(b);
    return r;
}
template <int L2, int V2, int L, int V>
BLSMI_DEV Fp2<L2, V2> fp2_relabel(const Fp2<L, V>& a) {
    Fp2<L2, V2> r;
    r.c0 = fp_relabel<L2, V2>(a.c0);
    r.c1 = fp_relabel<L2, V2>(a.c1);
    return r;
}
template <int L2, int V2, int L, int V>
BLSMI_DEV Fp6<L2, V2> fp6_relabel(const Fp6<L, V>& a) {
    Fp6<L2, V2> r;
    r.c0 = fp2_relabel<L2, V2>(a.c0);
    r.c1 = fp2_relabel<L2, V2>(a.c1);
    r.c2 = fp2_relabel<L2, V2>(a.c2);
    return r;
}
template <int La, int Va, int Lb, int Vb, int Lc, int Vc>
BLSMI_DEV auto make_fp6(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b, const Fp2<Lc, Vc>& c) {
    constexpr int L = imax(La, imax(Lb, Lc)), V = imax(Va, imax(Vb, Vc));
    Fp6<L, V> r;
    r.c0 = fp2_relabel<L, V>(a);
    r.c1 = fp2_relabel<L, V>(b);
    r.c2 = fp2_relabel<L, V>(c);
    return r;
}
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto make_fp12(const Fp6<La, Va>& a, const Fp6<Lb, Vb>& b) {
    constexpr int L = imax(La, Lb), V = imax(Va, Vb);
    Fp12<L, V> r;
    r.c0 = fp6_relabel<L, V>(a);
    r.c1 = fp6_relabel<L, V>(b);
    return r;
}

// ------------------------------------------------------------------------------------------------
// Fq2
// ------------------------------------------------------------------------------------------------
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto fp2_add(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return make_fp2(fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto fp2_sub(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return make_fp2(fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)); }
template <int L, int V> BLSMI_DEV auto fp2_neg(const Fp2<L, V>& a) { return make_fp2(fp_neg(a.c0), fp_neg(a.c1)); }
template <int L, int V> BLSMI_DEV auto fp2_dbl(const Fp2<L, V>& a) { return make_fp2(fp_dbl(a.c0), fp_dbl(a.c1)); }
template <int K, int L, int V> BLSMI_DEV auto fp2_muls(const Fp2<L, V>& a) { return make_fp2(fp_muls<K>(a.c0), fp_muls<K>(a.c1)); }
template <int L, int V> BLSMI_DEV auto fp2_conj(const Fp2<L, V>& a) { return make_fp2(a.c0, fp_neg(a.c1)); }
template <int L, int V> BLSMI_DEV auto fp2_norm(const Fp2<L, V>& a) { return make_fp2(fp_norm(a.c0), fp_norm(a.c1)); }
template <int L, int V> BLSMI_DEV Fp2S fp2_store(const Fp2<L, V>& a) { Fp2S r; r.c0 = fp_store(a.c0); r.c1 = fp_store(a.c1); return r; }
template <int L, int V> BLSMI_DEV bool fp2_is_zero(const Fp2<L, V>& a) { return fp_is_zero(a.c0) & fp_is_zero(a.c1); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV bool fp2_eq(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return fp2_is_zero(fp2_sub(a, b)); }
template <int L, int V> BLSMI_DEV Fp2<L, V> fp2_select(i32 m, const Fp2<L, V>& a, const Fp2<L, V>& b) { Fp2<L, V> r; r.c0 = fp_select(m, a.c0, b.c0); r.c1 = fp_select(m, a.c1, b.c1); return r; }
BLSMI_DEV Fp2S fp2_zero() { Fp2S r; r.c0 = fp_zero(); r.c1 = fp_zero(); return r; }
BLSMI_DEV Fp2S fp2_one() { Fp2S r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }

// fq2.go:41-45: multiply by the non-residue 1+u
template <int L, int V> BLSMI_DEV auto fp2_mul_nr(const Fp2<L, V>& a) { return make_fp2(fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)); }
// fq2.go:116-130 (Karatsuba, 3 Fq multiplications)
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto fp2_mul(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) {
    const auto aa = fp_mul(a.c0, b.c0);
    const auto bb = fp_mul(a.c1, b.c1);
    const auto t = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    return make_fp2(fp_sub(aa, bb), fp_sub(fp_sub(t, aa), bb));
}
// fq2.go:75-89 (complex squaring, 2 Fq multiplications)
template <int L, int V>
BLSMI_DEV auto fp2_sqr(const Fp2<L, V>& a) {
    const auto ab = fp_mul(a.c0, a.c1);
    const auto c0 = fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1));
    return make_fp2(c0, fp_dbl(ab));
}
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto fp2_mul_fp(const Fp2<La, Va>& a, const Fp<Lb, Vb>& s) { return make_fp2(fp_mul(a.c0, s), fp_mul(a.c1, s)); }
// fq2.go:133-147
template <int L, int V>
BLSMI_DEV auto fp2_inv(const Fp2<L, V>& a) {
    const FpS t = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
    return make_fp2(fp_mul(a.c0, t), fp_neg(fp_mul(a.c1, t)));
}
// fq2.go:156-158: c1 *= (-1)^power
template <int P, int L, int V>
BLSMI_DEV auto fp2_frob(const Fp2<L, V>& a) {
    if constexpr (P % 2 == 0) return a; else return fp2_conj(a);
}

// fq2.go:177-193: exponentiation by a fixed public exponent, bits MSB-first from constant memory
template <int L, int V>
BLSMI_DEV Fp2S fp2_pow_const(const Fp2<L, V>& a, const u32* ebits, int nbits) {
    const Fp2S base = fp2_store(a);
    Fp2S res = base;
    for (int i = nbits - 2; i >= 0; i--) {
        res = fp2_store(fp2_sqr(res));
        if ((ebits[i >> 5] >> (i & 31)) & 1) res = fp2_store(fp2_mul(res, base));
    }
    return res;
}
// fq2.go:198-232 (Algorithm 9 of eprint 2012/685), evaluated without divergent control flow:
// both tails are computed and the result is selected.
template <int L, int V>
BLSMI_DEV Fp2S fp2_sqrt(const Fp2<L, V>& a_in, bool& ok) {
    const Fp2S a = fp2_store(a_in);
    const bool zero = fp2_is_zero(a);
    const Fp2S a1 = fp2_pow_const(a, C_QM3O4, BLSMI_QM3O4_BITS);
    const Fp2S alpha = fp2_store(fp2_mul(fp2_sqr(a1), a));
    const Fp2S a0 = fp2_store(fp2_mul(fp2_conj(alpha), alpha));
    Fp2S neg1; neg1.c0 = C_NEGONE; neg1.c1 = fp_zero();
    const bool nonres = fp2_eq(a0, neg1);
    const Fp2S x0 = fp2_store(fp2_mul(a1, a));
    const bool alpha_m1 = fp2_eq(alpha, neg1);
    Fp2S xu; xu.c0 = fp_store(fp_neg(x0.c1)); xu.c1 = x0.c0;              // x0 * u
    const Fp2S b = fp2_pow_const(fp2_add(alpha, fp2_one()), C_QM1O2, BLSMI_QM1O2_BITS);
    const Fp2S xb = fp2_store(fp2_mul(b, x0));
    Fp2S r = fp2_select(alpha_m1 ? -1 : 0, xu, xb);
    r = fp2_select(zero ? -1 : 0, fp2_zero(), r);
    ok = zero | !nonres;
    return r;
}

// ------------------------------------------------------------------------------------------------
// Fq6
// ------------------------------------------------------------------------------------------------
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto fp6_add(const Fp6<La, Va>& a, const Fp6<Lb, Vb>& b) { return make_fp6(fp2_add(a.c0, b.c0), fp2_add(a.c1, b.c1), fp2_add(a.c2, b.c2)); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto fp6_sub(const Fp6<La, Va>& a, const Fp6<Lb, Vb>& b) { return make_fp6(fp2_sub(a.c0, b.c0), fp2_sub(a.c1, b.c1), fp2_sub(a.c2, b.c2)); }
template <int L, int V> BLSMI_DEV auto fp6_neg(const Fp6<L, V>& a) { return make_fp6(fp2_neg(a.c0), fp2_neg(a.c1), fp2_neg(a.c2)); }
template <int L, int V> BLSMI_DEV auto fp6_norm(const Fp6<L, V>& a) { return make_fp6(fp2_norm(a.c0), fp2_norm(a.c1), fp2_norm(a.c2)); }
template <int L, int V> BLSMI_DEV Fp6S fp6_store(const Fp6<L, V>& a) { Fp6S r; r.c0 = fp2_store(a.c0); r.c1 = fp2_store(a.c1); r.c2 = fp2_store(a.c2); return r; }
BLSMI_DEV Fp6S fp6_zero() { Fp6S r; r.c0 = fp2_zero(); r.c1 = fp2_zero(); r.c2 = fp2_zero(); return r; }
BLSMI_DEV Fp6S fp6_one() { Fp6S r; r.c0 = fp2_one(); r.c1 = fp2_zero(); r.c2 = fp2_zero(); return r; }
// fq6.go:34-37: multiply by v
template <int L, int V> BLSMI_DEV auto fp6_mul_nr(const Fp6<L, V>& a) { return make_fp6(fp2_mul_nr(a.c2), a.c0, a.c1); }
// fq6.go:255-292
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto fp6_mul(const Fp6<La, Va>& a, const Fp6<Lb, Vb>& b) {
    const auto aa = fp2_mul(a.c0, b.c0);
    const auto bb = fp2_mul(a.c1, b.c1);
    const auto cc = fp2_mul(a.c2, b.c2);
    const auto t1 = fp2_add(fp2_mul_nr(fp2_sub(fp2_sub(fp2_mul(fp2_add(b.c1, b.c2), fp2_add(a.c1, a.c2)), bb), cc)), aa);
    const auto t3 = fp2_sub(fp2_add(fp2_sub(fp2_mul(fp2_add(b.c0, b.c2), fp2_add(a.c0, a.c2)), aa), bb), cc);
    const auto t2 = fp2_add(fp2_sub(fp2_sub(fp2_mul(fp2_add(b.c0, b.c1), fp2_add(a.c0, a.c1)), aa), bb), fp2_mul_nr(cc));
    return make_fp6(t1, t2, t3);
}
// fq6.go:221-252
template <int L, int V>
BLSMI_DEV auto fp6_sqr(const Fp6<L, V>& a) {
    const auto s0 = fp2_sqr(a.c0);
    const auto s1 = fp2_dbl(fp2_mul(a.c0, a.c1));
    const auto s2 = fp2_sqr(fp2_add(fp2_sub(a.c0, a.c1), a.c2));
    const auto s3 = fp2_dbl(fp2_mul(a.c1, a.c2));
    const auto s4 = fp2_sqr(a.c2);
    const auto c0 = fp2_add(fp2_mul_nr(s3), s0);
    const auto c1 = fp2_add(fp2_mul_nr(s4), s1);
    const auto c2 = fp2_sub(fp2_sub(fp2_add(fp2_add(s1, s2), s3), s0), s4);
    return make_fp6(c0, c1, c2);
}
// fq6.go:40-57
template <int L, int V, int Lc, int Vc>
BLSMI_DEV auto fp6_mul_by_1(const Fp6<L, V>& a, const Fp2<Lc, Vc>& c1) {
    const auto b = fp2_mul(a.c1, c1);
    const auto t1 = fp2_mul_nr(fp2_sub(fp2_mul(c1, fp2_add(a.c1, a.c2)), b));
    const auto t2 = fp2_sub(fp2_mul(c1, fp2_add(a.c0, a.c1)), b);
    return make_fp6(t1, t2, b);
}
// fq6.go:60-90
template <int L, int V, int L0, int V0, int L1, int V1>
BLSMI_DEV auto fp6_mul_by_01(const Fp6<L, V>& a, const Fp2<L0, V0>& c0, const Fp2<L1, V1>& c1) {
    const auto aa = fp2_mul(a.c0, c0);
    const auto b = fp2_mul(a.c1, c1);
    const auto t1 = fp2_add(fp2_mul_nr(fp2_sub(fp2_mul(c1, fp2_add(a.c1, a.c2)), b)), aa);
    const auto t3 = fp2_add(fp2_sub(fp2_mul(c0, fp2_add(a.c0, a.c2)), aa), b);
    const auto t2 = fp2_sub(fp2_sub(fp2_mul(fp2_add(c0, c1), fp2_add(a.c0, a.c1)), aa), b);
    return make_fp6(t1, t2, t3);
}
// fq6.go:295-336
template <int L, int V>
BLSMI_DEV auto fp6_inv(const Fp6<L, V>& a) {
    const Fp2S c0 = fp2_store(fp2_add(fp2_neg(fp2_mul(fp2_mul_nr(a.c2), a.c1)), fp2_sqr(a.c0)));
    const Fp2S c1 = fp2_store(fp2_sub(fp2_mul_nr(fp2_sqr(a.c2)), fp2_mul(a.c0, a.c1)));
    const Fp2S c2 = fp2_store(fp2_sub(fp2_sqr(a.c1), fp2_mul(a.c0, a.c2)));
    auto t = fp2_mul_nr(fp2_add(fp2_mul(a.c2, c1), fp2_mul(a.c1, c2)));
    const Fp2S ti = fp2_store(fp2_inv(fp2_store(fp2_add(t, fp2_mul(a.c0, c0)))));
    return make_fp6(fp2_mul(ti, c0), fp2_mul(ti, c1), fp2_mul(ti, c2));
}
// fq6.go:211-218 with the coefficients (1+u)^((q^P-1)/3), (1+u)^((2q^P-2)/3) of consts.cuh
template <int P, int L, int V>
BLSMI_DEV auto fp6_frob(const Fp6<L, V>& a) {
    return make_fp6(fp2_frob<P>(a.c0), fp2_mul(fp2_frob<P>(a.c1), C_FROB6_C1[P % 6]), fp2_mul(fp2_frob<P>(a.c2), C_FROB6_C2[P % 6]));
}

// ------------------------------------------------------------------------------------------------
// Fq12
// ------------------------------------------------------------------------------------------------
template <int L, int V> BLSMI_DEV Fp12S fp12_store(const Fp12<L, V>& a) { Fp12S r; r.c0 = fp6_store(a.c0); r.c1 = fp6_store(a.c1); return r; }
BLSMI_DEV Fp12S fp12_one() { Fp12S r; r.c0 = fp6_one(); r.c1 = fp6_zero(); return r; }
template <int L, int V> BLSMI_DEV auto fp12_conj(const Fp12<L, V>& a) { return make_fp12(a.c0, fp6_neg(a.c1)); }   // fq12.go:27-29
// fq12.go:198-213
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto fp12_mul(const Fp12<La, Va>& a, const Fp12<Lb, Vb>& b) {
    const auto aa = fp6_norm(fp6_mul(a.c0, b.c0));       // tight bounds (normalised limbs, value bound kept)
    const auto bb = fp6_norm(fp6_mul(a.c1, b.c1));
    const auto t = fp6_mul(fp6_add(a.c1, a.c0), fp6_add(b.c0, b.c1));
    return make_fp12(fp6_add(fp6_mul_nr(bb), aa), fp6_sub(fp6_sub(t, aa), bb));
}
// fq12.go:180-195
template <int L, int V>
BLSMI_DEV auto fp12_sqr(const Fp12<L, V>& a) {
    const auto ab = fp6_norm(fp6_mul(a.c0, a.c1));
    const auto t = fp6_mul(fp6_add(fp6_mul_nr(a.c1), a.c0), fp6_add(a.c0, a.c1));
    return make_fp12(fp6_sub(fp6_sub(t, ab), fp6_mul_nr(ab)), fp6_add(ab, ab));
}
// fq12.go:32-47
template <int L, int V, int L0, int V0, int L1, int V1, int L4, int V4>
BLSMI_DEV auto fp12_mul_by_014(const Fp12<L, V>& a, const Fp2<L0, V0>& c0, const Fp2<L1, V1>& c1, const Fp2<L4, V4>& c4) {
    const auto aa = fp6_norm(fp6_mul_by_01(a.c0, c0, c1));
    const auto bb = fp6_norm(fp6_mul_by_1(a.c1, c4));
    const auto t = fp6_mul_by_01(fp6_add(a.c1, a.c0), c0, fp2_add(c1, c4));
    return make_fp12(fp6_add(fp6_mul_nr(bb), aa), fp6_sub(fp6_sub(t, aa), bb));
}
// fq12.go:216-237
template <int L, int V>
BLSMI_DEV auto fp12_inv(const Fp12<L, V>& a) {
    const Fp6S t = fp6_store(fp6_inv(fp6_store(fp6_sub(fp6_sqr(a.c0), fp6_mul_nr(fp6_store(fp6_sqr(a.c1)))))));
    return make_fp12(fp6_mul(t, a.c0), fp6_neg(fp6_mul(t, a.c1)));
}
// fq12.go:171-177 with (1+u)^((q^P-1)/6)
template <int P, int L, int V>
BLSMI_DEV auto fp12_frob(const Fp12<L, V>& a) {
    const auto c0 = fp6_frob<P>(a.c0);
    const auto c1 = fp6_frob<P>(a.c1);
    const Fp2S k = C_FROB12_C1[P % 12];
    return make_fp12(c0, make_fp6(fp2_mul(fp2_store(c1.c0), k), fp2_mul(fp2_store(c1.c1), k), fp2_mul(fp2_store(c1.c2), k)));
}
// Granger-Scott squaring for elements of the cyclotomic subgroup (after the easy part of the final
// exponentiation).  The reference has no such routine -- FQ12.Exp (fq12.go:108-120) squares with a
// full multiplication -- but on the subgroup the value is the same: x^2.
template <class A, class B> struct Pair { A first; B second; };
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV auto fp4_sqr(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) {
    const auto t0 = fp2_norm(fp2_sqr(a));
    const auto t1 = fp2_norm(fp2_sqr(b));
    const auto c0 = fp2_norm(fp2_add(fp2_mul_nr(t1), t0));
    const auto c1 = fp2_norm(fp2_sub(fp2_sub(fp2_sqr(fp2_add(a, b)), t0), t1));
    return Pair<decltype(c0), decltype(c1)>{c0, c1};
}
// outputs are 3t -+ 2z with z an input: the representation doubles per squaring, so each output is
// value-reduced (fp_store) -- the only place in the pairing where a reduction is inherent.
BLSMI_DEV Fp12S fp12_cyclotomic_sqr(const Fp12S& f) {
    const Fp2S z0 = f.c0.c0, z4 = f.c0.c1, z3 = f.c0.c2, z2 = f.c1.c0, z1 = f.c1.c1, z5 = f.c1.c2;
    Fp12S r;
    const auto a = fp4_sqr(z0, z1);
    r.c0.c0 = fp2_store(fp2_add(fp2_dbl(fp2_sub(a.first, z0)), a.first));
    r.c1.c1 = fp2_store(fp2_add(fp2_dbl(fp2_add(a.second, z1)), a.second));
    const auto b = fp4_sqr(z2, z3);
    const auto c = fp4_sqr(z4, z5);
    r.c0.c1 = fp2_store(fp2_add(fp2_dbl(fp2_sub(b.first, z4)), b.first));
    r.c1.c2 = fp2_store(fp2_add(fp2_dbl(fp2_add(b.second, z5)), b.second));
    const auto t3n = fp2_norm(fp2_mul_nr(c.second));
    r.c1.c0 = fp2_store(fp2_add(fp2_dbl(fp2_add(t3n, z2)), t3n));
    r.c0.c2 = fp2_store(fp2_add(fp2_dbl(fp2_sub(c.first, z3)), c.first));
    return r;
}
template <int La, int Va, int Lb, int Vb>
BLSMI_DEV bool fp12_eq(const Fp12<La, Va>& a, const Fp12<Lb, Vb>& b) {
    bool e = true;
    e &= fp2_eq(a.c0.c0, b.c0.c0); e &= fp2_eq(a.c0.c1, b.c0.c1); e &= fp2_eq(a.c0.c2, b.c0.c2);
    e &= fp2_eq(a.c1.c0, b.c1.c0); e &= fp2_eq(a.c1.c1, b.c1.c1); e &= fp2_eq(a.c1.c2, b.c1.c2);
    return e;
}

}  // namespace blsmi

// ------------------------------------------------------------------------------------------------
// Generic spellings so that the curve code (curve.cuh) is written once for Fq (G1) and Fq2 (G2)
// ------------------------------------------------------------------------------------------------
namespace blsmi {
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto f_add(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) { return fp_add(a, b); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto f_add(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return fp2_add(a, b); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto f_sub(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) { return fp_sub(a, b); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto f_sub(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return fp2_sub(a, b); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto f_mul(const Fp<La, Va>& a, const Fp<Lb, Vb>& b) { return fp_mul(a, b); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto f_mul(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return fp2_mul(a, b); }
template <int L, int V> BLSMI_DEV auto f_sqr(const Fp<L, V>& a) { return fp_sqr(a); }
template <int L, int V> BLSMI_DEV auto f_sqr(const Fp2<L, V>& a) { return fp2_sqr(a); }
template <int L, int V> BLSMI_DEV auto f_neg(const Fp<L, V>& a) { return fp_neg(a); }
template <int L, int V> BLSMI_DEV auto f_neg(const Fp2<L, V>& a) { return fp2_neg(a); }
template <int K, int L, int V> BLSMI_DEV auto f_muls(const Fp<L, V>& a) { return fp_muls<K>(a); }
template <int K, int L, int V> BLSMI_DEV auto f_muls(const Fp2<L, V>& a) { return fp2_muls<K>(a); }
template <int L, int V> BLSMI_DEV auto f_dbl(const Fp<L, V>& a) { return fp_muls<2>(a); }
template <int L, int V> BLSMI_DEV auto f_dbl(const Fp2<L, V>& a) { return fp2_muls<2>(a); }
template <int L, int V> BLSMI_DEV FpS f_store(const Fp<L, V>& a) { return fp_store(a); }
template <int L, int V> BLSMI_DEV Fp2S f_store(const Fp2<L, V>& a) { return fp2_store(a); }
template <int L, int V> BLSMI_DEV bool f_is_zero(const Fp<L, V>& a) { return fp_is_zero(a); }
template <int L, int V> BLSMI_DEV bool f_is_zero(const Fp2<L, V>& a) { return fp2_is_zero(a); }
template <int L, int V> BLSMI_DEV FpS f_inv(const Fp<L, V>& a) { return fp_inv(a); }
template <int L, int V> BLSMI_DEV Fp2S f_inv(const Fp2<L, V>& a) { return fp2_store(fp2_inv(a)); }
template <int L, int V> BLSMI_DEV Fp<L, V> f_select(i32 m, const Fp<L, V>& a, const Fp<L, V>& b) { return fp_select(m, a, b); }
template <int L, int V> BLSMI_DEV Fp2<L, V> f_select(i32 m, const Fp2<L, V>& a, const Fp2<L, V>& b) { return fp2_select(m, a, b); }
template <class F> struct field_consts;
template <> struct field_consts<FpS> { static BLSMI_DEV FpS zero() { return fp_zero(); } static BLSMI_DEV FpS one() { return fp_one(); } };
template <> struct field_consts<Fp2S> { static BLSMI_DEV Fp2S zero() { return fp2_zero(); } static BLSMI_DEV Fp2S one() { return fp2_one(); } };
}  // namespace blsmi
