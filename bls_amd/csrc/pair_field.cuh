// pair_field.cuh -- Fq2 in the lane-pair layout (fp2_pair.inc) as a FIELD for the generic curve code of curve.cuh:
// G2 Jacobian arithmetic with one Fq2 coefficient per lane, 45 registers of point state per lane instead of 90, two
// waves per SIMD.  Include AFTER every one-element-per-lane use of BLSMI_FP2_K in the translation unit (fp2_pair.inc
// re-points that macro at the lane-pair form of a constant).
#pragma once
#include "glv.cuh"        // (curve.cuh + the one-element-per-lane endomorphisms, which must see BLSMI_FP2_K before it is re-pointed)

#ifdef BLSMI_ASM_CORES
#ifdef BLSMI_LIMBS28
#include "core_asm28.inc"               // the same blobs for 14 x 28-bit limbs (gen_core_asm.py --limbs28)
#else
#include "core_asm.inc"
#endif
#endif
namespace blsmi {
namespace pairl {
#include "fp2_pair.inc"
// generic spellings (found by argument-dependent lookup from the templates of curve.cuh)
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto f_add(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return fp2_add(a, b); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto f_sub(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return fp2_sub(a, b); }
template <int La, int Va, int Lb, int Vb> BLSMI_DEV auto f_mul(const Fp2<La, Va>& a, const Fp2<Lb, Vb>& b) { return fp2_mul(a, b); }
template <int L, int V> BLSMI_DEV auto f_sqr(const Fp2<L, V>& a) { return fp2_sqr(a); }
template <int L, int V> BLSMI_DEV auto f_neg(const Fp2<L, V>& a) { return fp2_neg(a); }
template <int K, int L, int V> BLSMI_DEV auto f_muls(const Fp2<L, V>& a) { return fp2_muls<K>(a); }
template <int L, int V> BLSMI_DEV auto f_dbl(const Fp2<L, V>& a) { return fp2_muls<2>(a); }
template <int L, int V> BLSMI_DEV Fp2S f_store(const Fp2<L, V>& a) { return fp2_store(a); }
template <int L, int V> BLSMI_DEV bool f_is_zero(const Fp2<L, V>& a) { return fp2_is_zero(a); }
template <int L, int V> BLSMI_DEV Fp2S f_inv(const Fp2<L, V>& a) { return fp2_store(fp2_inv(a)); }
template <int L, int V> BLSMI_DEV Fp2<L, V> f_select(i32 m, const Fp2<L, V>& a, const Fp2<L, V>& b) { return fp2_select(m, a, b); }
using G2JacP = Jac<Fp2S>;
using G2AffP = Aff<Fp2S>;
#include "glv_endo2.inc"                                                  // psi, psi^2, psi^3 on lane-pair Jacobian points (glv.cuh finds them by argument-dependent lookup)
}  // namespace pairl
template <> struct glv_shape<pairl::Fp2S> { using type = GlvG2; };
template <> struct field_consts<pairl::Fp2S> { static BLSMI_DEV pairl::Fp2S zero() { return pairl::fp2_zero(); } static BLSMI_DEV pairl::Fp2S one() { return pairl::fp2_one(); } };
}  // namespace blsmi
