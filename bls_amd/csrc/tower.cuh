// tower.cuh -- Fq2 / Fq6 / Fq12 on the device (replaces fq2.go:41-232, fq6.go:34-336, fq12.go:27-237).
// Same tower as the reference: Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(1+u)), Fq12 = Fq6[w]/(w^2-v).
// Every function returns the mathematically same element as the cited reference method; bounds
// (<L,V>, see fp.cuh) are propagated by the types, so `auto` results are "lazy" and are brought back
// to the storage types (Fp2S/Fp6S/Fp12S) with *_store() where a value is kept across a loop.
#pragma once
#include "fp.cuh"
#include "fp_row.cuh"

namespace blsmi {

constexpr int imax(int a, int b) { return a > b ? a : b; }

#include "fp2_single.inc"
#include "tower_body.inc"

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
