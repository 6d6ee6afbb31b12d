// pairing.cuh -- optimal-ate Miller loop and final exponentiation, one (P, Q) tuple per lane.
//
// Reference path: G2AffineToPrepared (g2.go:650-801) stores 68 line-coefficient triples per Q
// (19.6 KB each), MillerLoop (pairing.go:16-75) replays them, FinalExponentiation
// (pairing.go:79-129) raises to 3(q^12-1)/r with a generic square-and-multiply.
// Here the line computation is FUSED into the Miller loop (no coefficient spill: the 68 triples
// never exist in memory), with the reference's exact doubling/addition formulas so that the
// Miller-loop output -- not only the final result -- is the same field element; and the hard part
// of the final exponentiation uses Granger-Scott cyclotomic squarings along the reference's
// addition chain, which yields the same power.
#pragma once
#include "curve.cuh"

namespace blsmi {

#define BLSMI_X_ABS 0xd201000000010000ULL          // |x|, blsIsNegative (g2.go:634-636)
#define BLSMI_NOINLINE __device__ __noinline__

struct G2Proj { Fp2S x, y, z; };

// Out-of-line Fq12 operations on references: between calls the 180-word operands live in the
// lane's scratch (a few hundred dword accesses against >10^4 VALU instructions per call), inside a
// call everything is in registers.  One copy of each body serves every kernel, which keeps each
// body's register allocation tractable and the code small.
BLSMI_NOINLINE void nf_fp12_mul(Fp12S& r, const Fp12S& a, const Fp12S& b) { r = fp12_store(fp12_mul(a, b)); }
BLSMI_NOINLINE void nf_fp12_sqr(Fp12S& r, const Fp12S& a) { r = fp12_store(fp12_sqr(a)); }
BLSMI_NOINLINE void nf_fp12_cyc_sqr(Fp12S& r, const Fp12S& a) { r = fp12_cyclotomic_sqr(a); }
BLSMI_NOINLINE void nf_fp12_inv(Fp12S& r, const Fp12S& a) { r = fp12_store(fp12_inv(a)); }
BLSMI_NOINLINE void nf_fp12_frob1(Fp12S& r, const Fp12S& a) { r = fp12_store(fp12_frob<1>(a)); }
BLSMI_NOINLINE void nf_fp12_frob2(Fp12S& r, const Fp12S& a) { r = fp12_store(fp12_frob<2>(a)); }
BLSMI_NOINLINE void nf_fp12_frob3(Fp12S& r, const Fp12S& a) { r = fp12_store(fp12_frob<3>(a)); }
BLSMI_DEV void fp12_conj_inplace(Fp12S& a) { a.c1 = fp6_store(fp6_neg(a.c1)); }

// g2.go:655-708
BLSMI_NOINLINE void doubling_step(G2Proj& r, Fp2S& o0, Fp2S& o1, Fp2S& o2) {
    const auto tmp0 = fp2_norm(fp2_sqr(r.x));
    const auto tmp1 = fp2_norm(fp2_sqr(r.y));
    const auto tmp2 = fp2_norm(fp2_sqr(tmp1));
    const auto tmp3 = fp2_norm(fp2_dbl(fp2_sub(fp2_sub(fp2_sqr(fp2_add(tmp1, r.x)), tmp0), tmp2)));
    const auto tmp4 = fp2_norm(fp2_muls<3>(tmp0));
    const auto tmp6 = fp2_norm(fp2_add(r.x, tmp4));
    const auto tmp5 = fp2_norm(fp2_sqr(tmp4));
    const auto zsq = fp2_norm(fp2_sqr(r.z));
    const auto nx = fp2_norm(fp2_sub(fp2_sub(tmp5, tmp3), tmp3));
    const auto nz = fp2_norm(fp2_sub(fp2_sub(fp2_sqr(fp2_add(r.z, r.y)), tmp1), zsq));
    const auto ny = fp2_norm(fp2_sub(fp2_mul(fp2_sub(tmp3, nx), tmp4), fp2_muls<8>(tmp2)));
    o1 = fp2_store(fp2_neg(fp2_dbl(fp2_mul(tmp4, zsq))));
    o2 = fp2_store(fp2_sub(fp2_sub(fp2_sub(fp2_sqr(tmp6), tmp0), tmp5), fp2_muls<4>(tmp1)));
    o0 = fp2_store(fp2_dbl(fp2_mul(nz, zsq)));
    r.x = fp2_store(nx); r.y = fp2_store(ny); r.z = fp2_store(nz);
}
// g2.go:710-772
BLSMI_NOINLINE void addition_step(G2Proj& r, const Fp2S& qx, const Fp2S& qy, Fp2S& o0, Fp2S& o1, Fp2S& o2) {
    const auto zsq = fp2_norm(fp2_sqr(r.z));
    const auto ysq = fp2_norm(fp2_sqr(qy));
    const auto t0 = fp2_norm(fp2_mul(zsq, qx));
    const auto t1 = fp2_norm(fp2_mul(fp2_sub(fp2_sub(fp2_sqr(fp2_add(qy, r.z)), ysq), zsq), zsq));
    const auto t2 = fp2_norm(fp2_sub(t0, r.x));
    const auto t3 = fp2_norm(fp2_sqr(t2));
    const auto t4 = fp2_norm(fp2_muls<4>(t3));
    const auto t5 = fp2_norm(fp2_mul(t4, t2));
    const auto t6 = fp2_norm(fp2_sub(fp2_sub(t1, r.y), r.y));
    const auto t9 = fp2_norm(fp2_mul(t6, qx));
    const auto t7 = fp2_norm(fp2_mul(t4, r.x));
    const auto nx = fp2_norm(fp2_sub(fp2_sub(fp2_sub(fp2_sqr(t6), t5), t7), t7));
    const auto nz = fp2_norm(fp2_sub(fp2_sub(fp2_sqr(fp2_add(r.z, t2)), zsq), t3));
    const auto t8 = fp2_norm(fp2_mul(fp2_sub(t7, nx), t6));
    const auto ny = fp2_norm(fp2_sub(t8, fp2_dbl(fp2_mul(r.y, t5))));
    const auto t10 = fp2_norm(fp2_sub(fp2_sub(fp2_sqr(fp2_add(qy, nz)), ysq), fp2_sqr(nz)));
    o2 = fp2_store(fp2_sub(fp2_dbl(t9), t10));
    o0 = fp2_store(fp2_dbl(nz));
    o1 = fp2_store(fp2_dbl(fp2_neg(t6)));
    r.x = fp2_store(nx); r.y = fp2_store(ny); r.z = fp2_store(nz);
}
// pairing.go:28-39: f *= line evaluated at P (sparse 014 multiplication)
BLSMI_NOINLINE void ell(Fp12S& f, const Fp2S& o0, const Fp2S& o1, const Fp2S& o2, const FpS& px, const FpS& py) {
    const auto c0 = fp2_mul_fp(o0, py);
    const auto c1 = fp2_mul_fp(o1, px);
    f = fp12_store(fp12_mul_by_014(f, o2, c1, c0));
}

// f = (f * line(P))^2 in one call: the 180-word accumulator crosses the call boundary once per iteration
BLSMI_NOINLINE void ell_sqr(Fp12S& f, const Fp2S& o0, const Fp2S& o1, const Fp2S& o2, const FpS& px, const FpS& py) {
    const auto c0 = fp2_mul_fp(o0, py);
    const auto c1 = fp2_mul_fp(o1, px);
    const Fp12S g = fp12_store(fp12_mul_by_014(f, o2, c1, c0));
    f = fp12_store(fp12_sqr(g));
}

// pairing.go:16-75 for NP pairs sharing the squarings; bits of |x|>>1 below its leading one.
template <int NP>
BLSMI_DEV void miller_loop(Fp12S& f, const G1Aff (&p)[NP], const G2Aff (&q)[NP]) {
    G2Proj r[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) { r[k].x = q[k].x; r[k].y = q[k].y; r[k].z = fp2_one(); }
    f = fp12_one();
    const u64 xr = BLSMI_X_ABS >> 1;
    Fp2S o0, o1, o2;
    for (int i = 61; i >= 0; i--) {
        const bool add = (xr >> i) & 1;
#pragma unroll
        for (int k = 0; k < NP; k++) {
            doubling_step(r[k], o0, o1, o2);
            if (!add && k == NP - 1) ell_sqr(f, o0, o1, o2, p[k].x, p[k].y);       // common case: multiply by the last line and square
            else ell(f, o0, o1, o2, p[k].x, p[k].y);
        }
        if (add) {
#pragma unroll
            for (int k = 0; k < NP; k++) {
                addition_step(r[k], q[k].x, q[k].y, o0, o1, o2);
                if (k == NP - 1) ell_sqr(f, o0, o1, o2, p[k].x, p[k].y);
                else ell(f, o0, o1, o2, p[k].x, p[k].y);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NP; k++) { doubling_step(r[k], o0, o1, o2); ell(f, o0, o1, o2, p[k].x, p[k].y); }
    fp12_conj_inplace(f);                                              // blsIsNegative (pairing.go:71-73)
}

// pairing.go:92-98 (ExpByX): f^|e| then conjugate.  f must lie in the cyclotomic subgroup.
BLSMI_NOINLINE void exp_by_x(Fp12S& out, const Fp12S& f, u64 e) {
    Fp12S res = f;
    for (int i = 62 - __builtin_clzll(e); i >= 0; i--) {
        res = fp12_cyclotomic_sqr(res);                                 // inlined: the accumulator stays in registers
        if ((e >> i) & 1) { Fp12S t = res; nf_fp12_mul(t, t, f); res = t; }   // rare (<= 6 set bits): spill only here
    }
    fp12_conj_inplace(res);
    out = res;
}
// pairing.go:79-129.  Computes r^(3(q^12-1)/r_order) in place; a non-invertible input (only 0)
// maps to 0 where the reference returns nil.
BLSMI_DEV void final_exponentiation(Fp12S& r) {
    Fp12S f2, y0, y1, y2, y3;
    nf_fp12_inv(f2, r);
    fp12_conj_inplace(r);
    nf_fp12_mul(r, r, f2);
    nf_fp12_frob2(f2, r);
    nf_fp12_mul(r, f2, r);
    const u64 x = BLSMI_X_ABS;
    nf_fp12_cyc_sqr(y0, r);
    exp_by_x(y1, y0, x);
    exp_by_x(y2, y1, x >> 1);
    y3 = r; fp12_conj_inplace(y3);
    nf_fp12_mul(y1, y1, y3);
    fp12_conj_inplace(y1);
    nf_fp12_mul(y1, y1, y2);
    exp_by_x(y2, y1, x);
    exp_by_x(y3, y2, x);
    fp12_conj_inplace(y1);
    nf_fp12_mul(y3, y3, y1);
    fp12_conj_inplace(y1);
    nf_fp12_frob3(y1, y1);
    nf_fp12_frob2(y2, y2);
    nf_fp12_mul(y1, y1, y2);
    exp_by_x(y2, y3, x);
    nf_fp12_mul(y2, y2, y0);
    nf_fp12_mul(y2, y2, r);
    nf_fp12_mul(y1, y1, y2);
    nf_fp12_frob1(y3, y3);
    nf_fp12_mul(r, y1, y3);
}

}  // namespace blsmi
