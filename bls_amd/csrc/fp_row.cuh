// fp_row.cuh -- one field element per WAVE (limb j in lane j) for chains of dependent multiplications that have nothing to
// run beside them: the 379-squaring exponentiation inside a square root when a single message is hashed (a lone Sign /
// Verify call).  One element per lane -- the layout of everything else -- makes such a chain 570 instructions per product in
// sequence; here the interleaved Montgomery product (R = 2^405 as everywhere, so values pass between the layouts untouched)
// is 15 steps of
//     P_k = a_i b_k + T_k           a_i broadcast with v_readlane, one v_mad_i64_i32 across the row
//     m   = -P_0 / q mod 2^27       lane 0 read back, two scalar instructions
//     P_k += m q_k
//     T_k = (P_{k+1} mod 2^27) + (P_k >> 27)
// -- dividing by 2^27 moves position k+1 to lane k (one DPP row shift folded into the add), and lane k's own excess belongs
// exactly there, so every limb stays a 32-bit value and nothing is carried along the row.  12 instructions a step; measured
// (tools/ubench_rowmul.hip) x^((q-3)/4): 219 us against 517 us, same field element.
// Preconditions: the wave is the workgroup (64 lanes, all active), operands wave-uniform.  Lanes 15..63 carry zeros.
#pragma once
#include "fp.cuh"
namespace blsmi {
BLSMI_DEV i32 row_mul(i32 a, i32 b, i32 qv) {
    i32 T = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const i32 sa = __builtin_amdgcn_readlane(a, i);
        i64 P = (i64)sa * b + (i64)T;
        const u32 t0 = (u32)__builtin_amdgcn_readlane((i32)P, 0);
        const i32 m = (i32)((t0 * BLSMI_QINV) & MASK);
        P += (i64)m * qv;
        const i32 lo = (i32)P & MASK;
        const i32 up = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, true);               // row_shl:1: lane k <- lane k+1
        T = up + (i32)(P >> LB);
    }
    return T;                                                              // limbs within a few units of [0, 2^27], the top one signed
}
// FOUR elements per wave (round 6): an element per DPP row of sixteen lanes, limb j in lane j of its row -- for calls of a few hundred to a few
// thousand messages, where a wave per element would be more waves than the chip holds and a lane per element leaves most SIMDs without one.
// Same steps; the two broadcasts are row_newbcast moves instead of v_readlane, m is formed by vector instructions (14 instructions a step).
template <int I>
BLSMI_DEV void row16_step(i32& T, i32 a, i32 b, i32 qv) {
    const i32 sa = __builtin_amdgcn_update_dpp(0, a, 0x150 + I, 0xf, 0xf, true);                 // row_newbcast:I
    i64 P = (i64)sa * b + (i64)T;
    const u32 t0 = (u32)__builtin_amdgcn_update_dpp(0, (i32)P, 0x150, 0xf, 0xf, true);           // row_newbcast:0
    const i32 m = (i32)((t0 * BLSMI_QINV) & MASK);
    P += (i64)m * qv;
    const i32 lo = (i32)P & MASK;
    const i32 up = __builtin_amdgcn_update_dpp(0, lo, 0x101, 0xf, 0xf, true);                   // row_shl:1: lane k <- lane k+1 of the row, 0 into lane 15
    T = up + (i32)(P >> LB);
    if constexpr (I + 1 < NL) row16_step<I + 1>(T, a, b, qv);
}
BLSMI_DEV i32 row16_mul(i32 a, i32 b, i32 qv) {
    i32 T = 0;
    row16_step<0>(T, a, b, qv);
    return T;
}
template <bool R16> BLSMI_DEV i32 row_mul_sel(i32 a, i32 b, i32 qv) { if constexpr (R16) return row16_mul(a, b, qv); else return row_mul(a, b, qv); }
// a^e, e given as in fp_pow_core (little-endian bit words, top bit set): 4-bit windows, the table one register per entry
template <bool R16>
BLSMI_DEV i32 row_pow_body(i32 a, const u32* ebits, int nbits, i32 qv) {
    i32 tab[16];
    tab[1] = a;
    for (int j = 2; j < 16; j++) tab[j] = (j & 1) ? row_mul_sel<R16>(tab[j - 1], a, qv) : row_mul_sel<R16>(tab[j >> 1], tab[j >> 1], qv);
    const int nw = (nbits + 3) >> 2;
    int i = nw - 1;
    i32 res = tab[(ebits[(4 * i) >> 5] >> ((4 * i) & 31)) & 15];
    for (i = nw - 2; i >= 0; i--) {
        for (int k = 0; k < 4; k++) res = row_mul_sel<R16>(res, res, qv);
        const u32 w = (ebits[(4 * i) >> 5] >> ((4 * i) & 31)) & 15;
        if (w) res = row_mul_sel<R16>(res, tab[w], qv);
    }
    return res;
}
__device__ __noinline__ i32 row_pow(i32 a, const u32* ebits, int nbits, i32 qv) { return row_pow_body<false>(a, ebits, nbits, qv); }
__device__ __noinline__ i32 row16_pow(i32 a, const u32* ebits, int nbits, i32 qv) { return row_pow_body<true>(a, ebits, nbits, qv); }
// the wave-uniform element x (every lane holds all of it) to the power e, back in every lane
template <int L, int V>
BLSMI_DEV FpS fp_pow_wave(const Fp<L, V>& x, const u32* ebits, int nbits) {
    const FpS base = fp_store(x);
    const int lane = threadIdx.x & 63;
    i32 a = 0, qv = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) { a = lane == i ? base.v[i] : a; qv = lane == i ? C_Q[i] : qv; }
    const i32 r = row_pow(a, ebits, nbits, qv);
    FpS y;
#pragma unroll
    for (int i = 0; i < NL; i++) y.v[i] = __builtin_amdgcn_readlane(r, i);
    return y;
}
template <int I>
BLSMI_DEV void row16_gather(FpS& y, i32 r) {
    y.v[I] = __builtin_amdgcn_update_dpp(0, r, 0x150 + I, 0xf, 0xf, true);
    if constexpr (I + 1 < NL) row16_gather<I + 1>(y, r);
}
// the ROW-uniform element x (every lane of a sixteen-lane row holds all of it; the four rows of a wave hold four different ones) to the power e
template <int L, int V>
BLSMI_DEV FpS fp_pow_row16(const Fp<L, V>& x, const u32* ebits, int nbits) {
    static_assert(NL <= 16, "a limb per lane of a DPP row");
    const FpS base = fp_store(x);
    const int lane = threadIdx.x & 15;
    i32 a = 0, qv = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) { a = lane == i ? base.v[i] : a; qv = lane == i ? C_Q[i] : qv; }
    const i32 r = row16_pow(a, ebits, nbits, qv);
    FpS y;
    row16_gather<0>(y, r);
    return y;
}
// SPREAD: 0 an element per lane (fp_pow_core), 1 an element per wave, 2 an element per row of sixteen lanes
template <int SPREAD, int L, int V>
BLSMI_DEV FpS fp_pow_spread(const Fp<L, V>& x, const u32* ebits, int nbits) {
    if constexpr (SPREAD == 2) return fp_pow_row16(x, ebits, nbits);
    else return fp_pow_wave(x, ebits, nbits);
}
}  // namespace blsmi
