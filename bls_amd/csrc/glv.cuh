// glv.cuh -- scalar multiplication through the curve endomorphisms (glv_model.py is the integer model of every step here).
//
// The reference multiplies bit-serially (g1.go:80-90, 562-585; g2.go:92-102, 609-632): one doubling per scalar bit.  For a point
// of the prime-order subgroup the same group element -- hence the same affine bytes -- is reached on a SHORTER doubling chain:
//   G1: phi(x, y) = (beta x, y) = [-z^2] (x, y), z = |x|.  k = k1 + k2 z^2  =>  [k] P = [k1] P + [k2] (-phi(P)): 2 x 129 bits.
//   G2: psi = [x] = [-z].   k = sum d_i z^i   =>  [k] P = [d0] P - [d1] psi(P) + [d2] psi^2(P) - [d3] psi^3(P): 4 x 65 bits.
// The ladder below walks Booth-recoded signed 5-bit digits of the sub-scalars over ONE table 0 P .. 16 P (the endomorphisms are
// applied to the entry picked, one or two multiplications by constants): G1 125 doublings + 52 additions + 16 for the table
// instead of 252 + 63 + 14, G2 65 + 56 + 16 instead of 252 + 63 + 14.
// Precondition: P in the subgroup (hash points, generators, deserialised keys and signatures are).  The host side keeps the
// plain windowed ladder for arbitrary curve points (blsmi_set_mul_assume_subgroup(0)).
#pragma once
#include "curve.cuh"

namespace blsmi {

// ---- multi-word integers (little-endian u32 words in registers) -------------------------------------------------------------
// out = (a * b) >> (32 DROP), NOUT words; every column from 0 is accumulated so that the carries into the kept part are exact
template <int NA, int NB, int DROP, int NOUT, class B>
BLSMI_DEV void umul_shift(const u32 (&a)[NA], const B& b, u32 (&out)[NOUT]) {
    u64 lo = 0; u32 hi = 0;
#pragma unroll
    for (int col = 0; col < DROP + NOUT; col++) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int j = col - i;
            if (j >= 0 && j < NB) { const u64 p = (u64)a[i] * (u32)b[j]; lo += p; hi += lo < p ? 1u : 0u; }
        }
        if (col >= DROP) out[col - DROP] = (u32)lo;
        lo = (lo >> 32) | ((u64)hi << 32); hi = 0;
    }
}
template <int N> BLSMI_DEV u32 usub(u32 (&a)[N], const u32 (&b)[N]) {          // a -= b, returns the borrow
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { const u64 d = (u64)a[i] - b[i] - br; a[i] = (u32)d; br = (u32)(d >> 63); }
    return br;
}
template <int N> BLSMI_DEV bool uge(const u32 (&a)[N], const u32 (&b)[N]) {    // a >= b
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { const u64 d = (u64)a[i] - b[i] - br; br = (u32)(d >> 63); }
    return br == 0;
}
template <int N> BLSMI_DEV void uinc(u32 (&a)[N]) {
    u32 c = 1;
#pragma unroll
    for (int i = 0; i < N; i++) { const u64 s = (u64)a[i] + c; a[i] = (u32)s; c = (u32)(s >> 32); }
}
// (q, r) = divmod(k, d) for an NK-word k and an ND-word divisor through the reciprocal M320 = floor(2^320 / d) (glv_model.div_recip):
// q^ = (k * (M320 >> 32 (8 - NK))) >> 32 (NK + 2) is q or q - 1; one correction.  q gets NQ words, r gets ND + 1.
template <int NK, int ND, int NM, int NQ, class D, class M>
BLSMI_DEV void udivmod_recip(const u32 (&k)[NK], const D& d, const M& m320, u32 (&q)[NQ], u32 (&r)[ND + 1]) {
    u32 m[NM - (8 - NK)];
#pragma unroll
    for (int i = 0; i < NM - (8 - NK); i++) m[i] = m320[i + 8 - NK];
    umul_shift<NK, NM - (8 - NK), NK + 2, NQ>(k, m, q);
    u32 t[ND + 1], dd[ND + 1];
    umul_shift<NQ, ND, 0, ND + 1>(q, d, t);                                // q d mod 2^(32 (ND + 1)): the remainder is below 2 d
#pragma unroll
    for (int i = 0; i <= ND; i++) { r[i] = i < NK ? k[i] : 0u; dd[i] = i < ND ? (u32)d[i] : 0u; }
    (void)usub(r, t);
    if (uge(r, dd)) { (void)usub(r, dd); uinc(q); }
}

// Booth digit i (5-bit windows) of the value whose DOUBLE is held in v2 (words in per-lane memory: the index is run-time):
// t = bits [5 i, 5 i + 6) of 2 v = bits [5 i - 1, 5 i + 5) of v;  digit = (t >> 1) + (t & 1) - 32 (t >> 5)  in [-16, 16]
template <int NW>
BLSMI_DEV i32 booth5(const u32 (&v2)[NW], int i) {
    const int bit = 5 * i, w = bit >> 5, sh = bit & 31;
    const u64 two = (u64)v2[w] | ((u64)(w + 1 < NW ? v2[w + 1] : 0u) << 32);
    const u32 t = (u32)(two >> sh) & 63u;
    return (i32)(t >> 1) + (i32)(t & 1u) - (i32)((t >> 5) << 5);
}
template <int N> BLSMI_DEV void ushl1(u32 (&a)[N]) {
#pragma unroll
    for (int i = N - 1; i > 0; i--) a[i] = (a[i] << 1) | (a[i - 1] >> 31);
    a[0] <<= 1;
}

// ---- the decompositions ------------------------------------------------------------------------------------------------------
// scalar: 32 big-endian bytes (FRRepr.Bytes, frrepr.go:188-195) -> 8 little-endian words
BLSMI_DEV void scalar_words(const u8* s, u32 (&k)[8]) {
    const u32* s32 = reinterpret_cast<const u32*>(s);
#pragma unroll
    for (int i = 0; i < 8; i++) k[i] = __builtin_bswap32(s32[7 - i]);
}
struct GlvG1 {
    static constexpr int NS = 2, NWIN = 26, KW = 5;                        // two sub-scalars below 2^129, doubled: 130 bits in 5 words
    static BLSMI_DEV void decompose(const u32 (&k)[8], u32 (&sub)[NS][KW]) {
        u32 k2[5], k1[5];
        udivmod_recip<8, 4, 7, 5>(k, C_GLV_Z2, C_GLV_MZ2, k2, k1);
#pragma unroll
        for (int i = 0; i < 5; i++) { sub[0][i] = k1[i]; sub[1][i] = k2[i]; }
        ushl1(sub[0]); ushl1(sub[1]);
    }
};
struct GlvG2 {
    static constexpr int NS = 4, NWIN = 14, KW = 3;                        // four digits below 2^65, doubled: 66 bits in 3 words
    static BLSMI_DEV void decompose(const u32 (&k)[8], u32 (&sub)[NS][KW]) {
        u32 q1[7], q2[5], q3[3], r[3];
        udivmod_recip<8, 2, 9, 7>(k, C_GLV_Z, C_GLV_MZ, q1, r);
#pragma unroll
        for (int i = 0; i < 3; i++) sub[0][i] = r[i];
        udivmod_recip<7, 2, 9, 5>(q1, C_GLV_Z, C_GLV_MZ, q2, r);
#pragma unroll
        for (int i = 0; i < 3; i++) sub[1][i] = r[i];
        udivmod_recip<5, 2, 9, 3>(q2, C_GLV_Z, C_GLV_MZ, q3, r);
#pragma unroll
        for (int i = 0; i < 3; i++) { sub[2][i] = r[i]; sub[3][i] = q3[i]; }
#pragma unroll
        for (int s = 0; s < 4; s++) ushl1(sub[s]);
    }
};

// ---- endomorphisms on Jacobian points: glv_endoS(P) = the base of the S-th summand, sign included ----------------------------
// G1: -phi(X, Y, Z) = (beta X, -Y, Z)
BLSMI_DEV Jac<FpS> glv_endo1(const Jac<FpS>& p) {
    Jac<FpS> r; r.x = fp_store(fp_mul(p.x, C_BETA)); r.y = fp_store(fp_neg(p.y)); r.z = p.z; r.inf = p.inf; return r;
}
#include "glv_endo2.inc"                                                  // G2, one element per lane (pair_field.cuh: the lane-pair twin)
template <class F> struct glv_shape;
template <> struct glv_shape<FpS> { using type = GlvG1; };
template <> struct glv_shape<Fp2S> { using type = GlvG2; };

// endo_s(e), s = the stream: run-time form of glv_endo1/2/3 for the rolled loop of the ladder (one copy of the addition body).
// G1: stream 1 = -phi: (beta X, -Y, Z)
BLSMI_DEV Jac<FpS> glv_endo_s(const Jac<FpS>& p, int s) {
    const i32 m = -(i32)(s & 1);
    Jac<FpS> r;
    r.x = fp_select(m, fp_store(fp_mul(p.x, C_BETA)), p.x);
    r.y = fp_select(m, fp_store(fp_neg(p.y)), p.y);
    r.z = p.z; r.inf = p.inf;
    return r;
}
// [k] P for P in the subgroup: signed-window ladder over the sub-scalars, one shared table.  F = FpS (G1), Fp2S (G2, either layout).
// The accumulator stays in registers for the whole ladder: the doubling and addition bodies are inlined ONCE each (rolled inner
// loops over the five doublings and over the streams), only the field cores are calls.
template <class F>
__device__ Jac<F> glv_mul(const Aff<F>& p, const u8* scalar) {
    using G = typename glv_shape<F>::type;
    u32 k[8];
    scalar_words(scalar, k);
    u32 sub[G::NS][G::KW];
    G::decompose(k, sub);
    // the table [1..16] P in AFFINE coordinates -- one shared inversion (Montgomery's trick over the 15 z's) -- so that the 52
    // additions of the ladder are mixed additions (7M + 4S instead of 11M + 5S); an infinite P makes every entry infinite
    // (the products vanish, inverse(0) = 0, the flags say so)
    Jac<F> tj[17];
    tj[1] = to_jac(p);
    for (int j = 2; j <= 16; j++) tj[j] = (j & 1) ? jac_add_affine(tj[j - 1], p) : jac_double(tj[j >> 1]);   // a doubling is cheaper than a mixed addition
    F pre[17];
    pre[2] = tj[2].z;
    for (int j = 3; j <= 16; j++) pre[j] = f_store(f_mul(pre[j - 1], tj[j].z));
    F inv = f_store(f_inv(pre[16]));                                        // 1 / (z_2 ... z_16)
    tj[0] = tj[1]; tj[0].inf = -1;                                          // digit 0: nothing is added
    for (int j = 16; j >= 2; j--) {                                         // in place: (x, y) become affine, z is not read again
        const F zi = j > 2 ? f_store(f_mul(inv, pre[j - 1])) : inv;         // 1 / z_j
        if (j > 2) inv = f_store(f_mul(inv, tj[j].z));
        const F zi2 = f_store(f_sqr(zi));
        tj[j].x = f_store(f_mul(tj[j].x, zi2));
        tj[j].y = f_store(f_mul(f_mul(tj[j].y, zi2), zi));
    }
    // (the accumulator is a local of its own, copied into the returned object at the end: as the returned object itself -- NRVO -- it is the
    //  caller's memory slot, and every doubling stored it there)
    Jac<F> acc = jac_zero<F>();
#pragma unroll 1
    for (int w = G::NWIN - 1; w >= 0; w--) {
        if (w != G::NWIN - 1) {
#pragma unroll 1
            for (int d = 0; d < 5; d++) acc = jac_double_i(acc);
        }
#pragma unroll 1
        for (int s = 0; s < G::NS; s++) {
            const i32 d = booth5(sub[s], w);
            const i32 neg = d >> 31;                                        // all-ones for a negative digit
            Jac<F> t = tj[(d ^ neg) - neg];
            t.z = tj[1].z;                                                  // = 1, and stays 1 under every endomorphism
            const Jac<F> e = glv_endo_s(t, s);
            Aff<F> ea; ea.x = e.x; ea.y = f_select(neg, f_store(f_neg(e.y)), e.y); ea.inf = e.inf;
            acc = jac_add_affine_i(acc, ea);
        }
    }
    Jac<F> res;
    res.x = acc.x; res.y = acc.y; res.z = acc.z; res.inf = acc.inf;
    return res;
}

}  // namespace blsmi
