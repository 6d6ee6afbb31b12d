// k_hash_pair.hip -- HashG2 (hash.go:391-411) for large batches with a LANE PAIR per message, two waves per SIMD.
// The one-lane k_hash_g2 (k_hash.hip) holds whole Fq2 values per lane: 256 registers + 256 accumulation registers of spill space and
// 5.3 KB of scratch, one wave per SIMD, one VALU instruction per 5.6 cycles.  Here the Fq2 arithmetic of the two SWU maps
// (g2.go:933-1031), of the 3-isogeny (hash.go:282-303) and of clearH2 (hash.go:341-389) runs in the lane-pair layout of the pairing
// kernels (fp2_pair.inc: one coefficient per lane, fused two-product multiply), and the Fq-only chains in between -- the norm root and
// the root of each map, one 379-squaring exponentiation each -- are split over the pair: the even lane takes map 1's, the odd lane
// map 2's, so no lane repeats its partner's work.  Same point, same bytes as hash_g2.  Inputs the fused identities exclude
// (g(x0) = 0, the two maps landing on opposite points) are flagged in good[] and redone by k_hash_g2_redo.
#ifndef BLSMI_LIMBS28                // (with the 14-limb blobs clang 22's machine scheduler segfaults on iso3_jac: function cores in that build)
#define BLSMI_ASM_CORES          // the lane-pair multiply cores as assembly blobs (core_asm.inc), as in k_fe_pair.hip: 6.41 -> 6.23 ms per 65 536 messages
#endif
#include "hash.cuh"
#include "device_io.cuh"
#include "pair_field.cuh"
using namespace blsmi;
namespace P2 = blsmi::pairl;
#include "pair_point_io.inc"

namespace blsmi {
namespace pairl {
BLSMI_DEV i32 dpp_odd(i32 x) { return __builtin_amdgcn_update_dpp(0, x, 0xF5, 0xf, 0xf, true); }     // quad_perm [1,1,3,3]: the pair's odd lane on both lanes
BLSMI_DEV FpS fp_of_even(const FpS& a) { FpS r; for (int i = 0; i < NL; i++) r.v[i] = dpp_even(a.v[i]); return r; }
BLSMI_DEV FpS fp_of_odd(const FpS& a) { FpS r; for (int i = 0; i < NL; i++) r.v[i] = dpp_odd(a.v[i]); return r; }
BLSMI_DEV FpS fp2_norm_fq(const Fp2S& a) {                                  // c0^2 + c1^2 on both lanes
    const FpS n = fp_store(fp_sqr(a.c));
    return fp_store(fp_add(n, fp_partner(n)));
}
// what a map's Fq2 half leaves for its Fq-only half and for the second Fq2 half
struct SwuState { Fp2S t, num, den, U, V; FpS nden, aa, nt; };

// the Fq2 part of OptimizedSWU2MapHelper before the first exponentiation (hash.cuh: swu_g2_helper_t), both lanes of the pair
__device__ __noinline__ void swu2_pair_head(SwuState& st, const Fp2S& t) {
    const Fp2S tsq = fp2_store(fp2_sqr(t));
    const Fp2S nqr_tsq = fp2_store(fp2_mul_nr(tsq));
    const Fp2S ndc = fp2_store(fp2_add(fp2_sqr(nqr_tsq), nqr_tsq));       // nqr^2 t^4 + nqr t^2
    const i32 ndc0 = fp2_is_zero(ndc) ? -1 : 0;
    const Fp2S B = BLSMI_FP2_K(C_ELL2PB), A = BLSMI_FP2_K(C_ELL2PA);
    st.num = fp2_select(ndc0, B, fp2_store(fp2_mul(fp2_neg(B), fp2_add(ndc, fp2_one()))));
    st.den = fp2_select(ndc0, fp2_store(fp2_mul_nr(A)), fp2_store(fp2_mul(A, ndc)));
    const Fp2S den2 = fp2_store(fp2_sqr(st.den));
    st.V = fp2_store(fp2_mul(den2, st.den));
    const Fp2S nd2 = fp2_store(fp2_mul(st.num, den2));
    st.U = fp2_store(fp2_add(fp2_add(fp2_mul(fp2_sqr(st.num), st.num), fp2_mul(A, nd2)), fp2_mul(B, st.V)));
    st.t = t;
    st.nden = fp2_norm_fq(st.den);
    st.aa = fp2_norm_fq(st.U);
    st.nt = fp2_norm_fq(t);
}
// the first Fq-only chain, one map per LANE: s0 = the norm root of g(x0) (or of -g(x0)), 1/b, N(den)^2/b
struct SwuMid { FpS s0, binv, k1; i32 m0, special; };
__device__ __noinline__ void swu2_lane_mid(SwuMid& o, const FpS& nden, const FpS& aa) {
    const FpS nden2 = fp_store(fp_sqr(nden));
    const FpS bb = fp_store(fp_mul(nden2, nden));                           // b = N(V) = N(den)^3
    const FpS ab = fp_store(fp_mul(aa, bb)), b2 = fp_store(fp_sqr(bb));
    const FpS e = fp_pow_qm3o4(fp_mul(ab, b2));
    o.s0 = fp_store(fp_mul(ab, e));
    o.m0 = fp_eq(fp_mul(fp_sqr(o.s0), bb), aa) ? -1 : 0;                   // N(g(x0)) is a square <=> g(x0) is a square
    const FpS binv_p = fp_store(fp_mul(fp_mul(ab, bb), fp_sqr(e)));        // chi / b
    o.binv = fp_select(o.m0, binv_p, fp_store(fp_neg(binv_p)));
    o.k1 = fp_store(fp_mul(nden2, o.binv));
    o.special = fp_is_zero(aa) ? -1 : 0;
}
// the Fq2 part between the exponentiations: x and the g(x) whose root is wanted
__device__ __noinline__ void swu2_pair_select(Fp2S& x, Fp2S& g, const SwuState& st, const FpS& k1, const FpS& binv, i32 m0) {
    const Fp2S x0 = fp2_store(fp2_mul_fp(fp2_mul(st.num, fp2_conj(st.den)), k1));
    const Fp2S gx0 = fp2_store(fp2_mul_fp(fp2_mul(st.U, fp2_conj(st.V)), binv));
    const Fp2S tsq = fp2_store(fp2_sqr(st.t));
    const Fp2S nqr_tsq = fp2_store(fp2_mul_nr(tsq));
    const Fp2S x1 = fp2_store(fp2_mul(nqr_tsq, x0));
    const Fp2S t6 = fp2_store(fp2_sqr(fp2_mul(tsq, st.t)));
    const Fp2S nqr3 = fp2_store(fp2_mul_nr(fp2_mul_nr(wrap(FpS(C_ONE)))));  // (1 + u)^3
    const Fp2S gx1 = fp2_store(fp2_mul(fp2_mul(nqr3, t6), gx0));          // g2.go:1005-1010
    x = fp2_select(m0, x0, x1);
    g = fp2_select(m0, gx0, gx1);
}
// the second Fq-only chain, one map per lane: the root from its norm root, with the reference's sign (g2.go:983-988, 1021-1026)
__device__ __noinline__ void swu2_lane_root(blsmi::Fp2S& y, const blsmi::Fp2S& g, const blsmi::Fp2S& t, const FpS& nt, const FpS& s0, i32 m0) {
    const FpS s1 = fp_store(fp_mul(fp_mul(fp_mul(fp_sqr(nt), nt), s0), C_SQRT_M8));   // N(gx1) = 8 N(t)^6 N(gx0) = s1^2
    y = blsmi::fp2_sqrt_from_norm_root(g, fp_select(m0, s0, s1));
    const i32 flip = fp2_sign_is_neg(t) ^ fp2_sign_is_neg(y);
    y = blsmi::fp2_select(flip, blsmi::fp2_store(blsmi::fp2_neg(y)), y);
}
// both coefficients of the pair's value v[m], m = this lane's map, on this lane (c0 from the even lane, c1 from the odd lane)
BLSMI_DEV blsmi::Fp2S gather_own_map(const Fp2S& v0, const Fp2S& v1, i32 odd) {
    const FpS recv = fp_partner(fp_select(odd, v0.c, v1.c));               // even lane sends v1.c0, odd lane sends v0.c1
    blsmi::Fp2S r;
    r.c0 = fp_select(odd, recv, v0.c);
    r.c1 = fp_select(odd, v1.c, recv);
    return r;
}
// and back: lane m holds y_m whole; the pair's y_0 and y_1 in the pair layout
BLSMI_DEV void scatter_own_map(Fp2S& y0, Fp2S& y1, const blsmi::Fp2S& y, i32 odd) {
    const FpS recv = fp_partner(fp_select(odd, y.c0, y.c1));               // even lane sends y_0.c1, odd lane sends y_1.c0
    y0.c = fp_select(odd, recv, y.c0);
    y1.c = fp_select(odd, y.c1, recv);
}
// iso3 on a Jacobian point (hash.cuh: iso_jac), the coefficient tables read per lane
__device__ __noinline__ void iso3_jac(G2JacP& out, const G2JacP& p) {
    constexpr int D = 3;
    Fp2S wp[D + 1];
    wp[1] = fp2_store(fp2_sqr(p.z));
    for (int i = 2; i <= D; i++) wp[i] = fp2_store(fp2_mul(wp[i - 1], wp[1]));
    auto hom = [&](const blsmi::Fp2S* c, int d) {
        Fp2S v = BLSMI_FP2_K(c[d]);
        for (int i = d - 1; i >= 0; i--) v = fp2_store(fp2_add(fp2_mul(v, p.x), fp2_mul(BLSMI_FP2_K(c[i]), wp[d - i])));
        return v;
    };
    const Fp2S XN = hom(C_XNUM3, 3), XD = hom(C_XDEN3, 2), YN = hom(C_YNUM3, 3), YD = hom(C_YDEN3, 3);
    const Fp2S xdyd = fp2_store(fp2_mul(XD, YD));
    const Fp2S yd2 = fp2_store(fp2_sqr(YD)), xd2 = fp2_store(fp2_sqr(XD));
    out.z = fp2_store(fp2_mul(p.z, xdyd));
    out.x = fp2_store(fp2_mul(fp2_mul(XN, XD), yd2));
    out.y = fp2_store(fp2_mul(fp2_mul(fp2_mul(p.y, YN), fp2_mul(xd2, XD)), yd2));
    out.inf = p.inf | (fp2_is_zero(out.z) ? -1 : 0);
}
// clearH2 (hash.go:368-389) on a Jacobian point: [x^2 - x - 1] P + [x - 1] psi(P) + psi^2(2 P), x = -|x|
BLSMI_DEV G2JacP clear_h2_jac_j(const G2JacP& p) {
    G2JacP work = jac_mul_u64_public(p, BLSMI_X_ABS);
    work = jac_add(work, p);
    const G2JacP mpsi = glv_endo1(p);                                      // -psi(P)
    work = jac_add(work, mpsi);
    work = jac_mul_u64_public(work, BLSMI_X_ABS);
    work = jac_add(work, mpsi);
    work = jac_add(work, jac_neg(p));
    return jac_add(work, glv_endo2(jac_double(p)));                        // psi^2(2 P)
}
__device__ __noinline__ void clear_h2_jac(G2AffP& out, const G2JacP& p) { out = jac_to_affine(clear_h2_jac_j(p)); }
// ScaleByCofactor (g2.go:104-115, 130-138) as hash.cuh's scale_by_cofactor_g2: [h2] P = [d_0](Q_0 + 2 Q_1 + 2 Q_2 + Q_3) - (Q_1 + 2 Q_2 + Q_3),
// Q_i = (-1)^i psi^i(clearH2(P)), d_0 = (|x| + 1) / 3
__device__ __noinline__ void scale_by_cofactor(G2AffP& out, const G2AffP& pt) {
    const G2JacP q0 = clear_h2_jac_j(to_jac(pt));
    const G2JacP q1 = glv_endo1(q0), q2 = glv_endo2(q0), q3 = glv_endo3(q0);
    const G2JacP a = jac_add(q1, q2);
    const G2JacP t = jac_add(jac_add(q0, q3), jac_double(a));
    const G2JacP s = jac_add(jac_add(a, q2), q3);
    out = jac_to_affine(jac_add(jac_mul_h2_d0(t), jac_neg(s)));
}
}  // namespace pairl
}  // namespace blsmi

// everything of a message's hash up to the 3-isogeny's Jacobian image (both kernels below); `bad`: the fused identities do not cover this message
BLSMI_DEV void hash_g2_pair_front(const u8* msgs, const u64* off, size_t n, unsigned redo_every, size_t& t0, size_t& t, int& par, P2::G2JacP& ij, i32& bad) {
    par = threadIdx.x & 1;
    const i32 odd = -(i32)par;
    t0 = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    t = t0 < n ? t0 : n - 1;                                               // both lanes of a pair stay active
    u32 d[8];
    sha256_msg(d, 1, 0x01, msgs + off[t], (size_t)(off[t + 1] - off[t]));
    // hp2 (hash.go:74-113): this lane's coefficient of t1 and of t2
    P2::SwuState st[2];
    for (int m = 0; m < 2; m++) {
        u32 d1[8], d2[8];
        sha256_35(d1, d, (u32)m, (u32)par + 1, 1); sha256_35(d2, d, (u32)m, (u32)par + 1, 2);
        P2::swu2_pair_head(st[m], P2::wrap(fq_from_two_digests(d1, d2)));
    }
    P2::SwuMid mid;
    P2::swu2_lane_mid(mid, fp_select(odd, st[1].nden, st[0].nden), fp_select(odd, st[1].aa, st[0].aa));
    const i32 special = mid.special | __shfl_xor(mid.special, 1);
    P2::Fp2S x[2], g[2];
    {
        const i32 m0_0 = __shfl(mid.m0, (int)(threadIdx.x & ~1u)), m0_1 = __shfl(mid.m0, (int)(threadIdx.x | 1u));
        P2::swu2_pair_select(x[0], g[0], st[0], P2::fp_of_even(mid.k1), P2::fp_of_even(mid.binv), m0_0);
        P2::swu2_pair_select(x[1], g[1], st[1], P2::fp_of_odd(mid.k1), P2::fp_of_odd(mid.binv), m0_1);
    }
    blsmi::Fp2S y;
    P2::swu2_lane_root(y, P2::gather_own_map(g[0], g[1], odd), P2::gather_own_map(st[0].t, st[1].t, odd),
                       fp_select(odd, st[1].nt, st[0].nt), mid.s0, mid.m0);
    P2::G2AffP a1, a2;
    a1.x = x[0]; a2.x = x[1]; a1.inf = 0; a2.inf = 0;
    P2::scatter_own_map(a1.y, a2.y, y, odd);
    // hash.go:391-411: the sum of the two mapped points, the 3-isogeny, clearH2 -- Jacobian throughout, one inversion
    const P2::G2JacP sj = jac_add_affine(to_jac(a1), a2);
    P2::iso3_jac(ij, sj);
    // redo_every (tests only): hand every redo_every-th message to the one-lane routine as if it had been exceptional
    bad = special | sj.inf | ((redo_every && t % redo_every == 0) ? -1 : 0);
}
__global__ void __launch_bounds__(WG, 2) k_hash_g2_pair(const u8* msgs, const u64* off, u8* good, u8* out, size_t n, unsigned redo_every) {
    hash_prio();
    size_t t0, t; int par; P2::G2JacP ij; i32 bad;
    hash_g2_pair_front(msgs, off, n, redo_every, t0, t, par, ij, bad);
    P2::G2AffP r; P2::clear_h2_jac(r, ij);
    if (t0 < n) {
        if (!par) good[t] = bad ? 0 : 1;
        if (!bad) {                                                        // x.c_par at +48 par, y.c_par at +96 + 48 par
            u8* o = out + 192 * t;
            if (r.inf) { u32* w = reinterpret_cast<u32*>(o + 48 * par); for (int i = 0; i < 12; i++) { w[i] = 0; w[24 + i] = 0; } }
            else { store_be48(o + 48 * par, r.x.c); store_be48(o + 96 + 48 * par, r.y.c); }
        }
    }
}

// the same up to the isogeny, for the row layout's tail (k_pairing_row.hip: k_clear_h2_row): the Jacobian image into `jbuf` (structure of arrays, element
// 2 c + parity of coordinate c), good[t] = 0 also when the image is the point at infinity
__global__ void __launch_bounds__(WG, 2) k_hash_g2_front(const u8* msgs, const u64* off, u8* good, i32* jbuf, size_t n, unsigned redo_every) {
    hash_prio();
    size_t t0, t; int par; P2::G2JacP ij; i32 bad;
    hash_g2_pair_front(msgs, off, n, redo_every, t0, t, par, ij, bad);
    bad |= ij.inf;
    if (t0 < n) {
        if (!par) good[t] = bad ? 0 : 1;
        soa_store(jbuf, n, t, 0 + par, ij.x.c); soa_store(jbuf, n, t, 2 + par, ij.y.c); soa_store(jbuf, n, t, 4 + par, ij.z.c);
    }
}

// HashG2WithDomain of a large batch, second half: ScaleByCofactor of the points k_tai_g2_wave (k_hash.hip) left, a lane pair per point
__global__ void __launch_bounds__(WG, 2) k_cofac2_pair(const u8* pts, u8* out, size_t n) {
    const int par = threadIdx.x & 1;
    const size_t t0 = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t t = t0 < n ? t0 : n - 1;
    P2::G2AffP r;
    P2::scale_by_cofactor(r, pair_load_g2(pts + 192 * t, par));
    if (t0 < n) pair_store_g2(out + 192 * t, par, r);
}
