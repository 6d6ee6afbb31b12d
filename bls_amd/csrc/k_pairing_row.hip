// k_pairing_row.hip -- pairing kernels in the LANE-ROW layout (row_body.inc): sixteen adjacent lanes per tuple, 4 tuples per 64-lane
// workgroup.  For the call sizes the reference's API produces -- a few thousand tuples: 4 096 tuples are 1 024 waves here, one on every
// SIMD of the chip (lane quad: 256 waves; one tuple per wave: 4 096 waves of 3.3 x the lane-instructions).  A lane holds one Fq of an
// Fq12 (14 words), so the kernels fit the 256-register budget of two waves per SIMD and serve up to ~10 000 tuples.
#ifndef BLSMI_ROW_WAVES
#define BLSMI_ROW_WAVES 2
#endif
// the Fq2 product core expanded in line at its call sites (fp2_pair.inc): the row routines are short sequences around one to three products, a call
// costs ~45 argument / result moves each.  Same box, interleaved (tools/ab_bench.py): 4 096 pairings 2.46 -> 2.38 ms, 8 192: 4.14 -> 4.06
#ifndef BLSMI_PAIR_CORE_CALL
#define BLSMI_PAIR_CORE_INLINE
#endif
#include "pairing.cuh"
#include "device_io.cuh"
namespace blsmi {
namespace pairl {
#include "row_body.inc"
#include "row_g2.inc"
#include "oct_g2.inc"
}  // namespace pairl
}  // namespace blsmi

#define KERNEL_ROW __global__ void __launch_bounds__(WG, BLSMI_ROW_WAVES)
namespace P2 = blsmi::pairl;
constexpr int RT = WG / 16;                                              // tuples per workgroup

// the hand-off buffer between Miller loop and final exponentiation is the lane-pair kernels' (device_io.cuh: soa_store): Fq number
// e = 2 (3 half + j) + parity of tuple t.  Pair p of a row holds the Fq2 coefficient j = p >> 1 of half p & 1.
BLSMI_DEV int row_fq_index(int pr, int par) { return 2 * (3 * (pr & 1) + (pr >> 1)) + par; }
BLSMI_DEV void row_store12(i32* buf, size_t n, size_t t, int pr, int par, const P2::R12& f) {
    if (pr < 6) soa_store(buf, n, t, row_fq_index(pr, par), fp_relabel<FpS::L, FpS::V>(f.c.c));
}
BLSMI_DEV P2::R12 row_load12(const i32* buf, size_t n, size_t t, int pr, int par) {
    P2::R12 f;
    f.c = P2::fp2_tight(P2::wrap(soa_load(buf, n, t, row_fq_index(pr < 6 ? pr : 0, par))));
    return f;
}
// Pairing(P, Q) = FinalExponentiation(MillerLoop) (pairing.go:132-136): the Miller value by the homogeneous steps, as k_miller1h_pair
KERNEL_ROW k_miller1h_row(const u8* g1, const u8* g2, i32* fbuf, size_t n) {
    const int par = threadIdx.x & 1, pr = (threadIdx.x >> 1) & 7;
    const size_t t = (size_t)blockIdx.x * RT + (threadIdx.x >> 4);
    const size_t tt = t < n ? t : n - 1;                                  // all sixteen lanes of a row stay active (DPP exchanges)
    const FpS px = load_be48(g1 + 96 * tt), py = load_be48(g1 + 96 * tt + 48);
    P2::R12 f;
    P2::miller_loop_r(f, px, py, P2::RQ{g2 + 192 * tt, par});
    if (t < n) row_store12(fbuf, n, t, pr, par, f);
}
// pairing.go:79-129 on the hand-off buffer; out = the reference's in-memory FQ12 (72 u64 per tuple).  mode 1: no exponentiation
KERNEL_ROW k_final_exp_row(const i32* fbuf, u64* out, size_t n, int mode) {
    const int par = threadIdx.x & 1, pr = (threadIdx.x >> 1) & 7;
    const size_t t = (size_t)blockIdx.x * RT + (threadIdx.x >> 4);
    const size_t tt = t < n ? t : n - 1;
    P2::R12 f = row_load12(fbuf, n, tt, pr, par);
    if (mode == 0) P2::final_exponentiation_r(f);
    if (t < n && pr < 6) store_m384(out + 72 * t + 6 * row_fq_index(pr, par), f.c.c);
}
// CompareTwoPairings in the row layout (k_miller2_pair's arguments: strides in bytes, 0 = one broadcast record; P1 is negated here;
// `pre`: pair 0's G2 point is the generator, read its prepared lines)
KERNEL_ROW k_miller2_row(const u8* p0, size_t sp0, const u8* q0, size_t sq0, const u8* p1, size_t sp1, const u8* q1, size_t sq1, i32* fbuf, size_t n, const i32* pre) {
    const int par = threadIdx.x & 1, pr = (threadIdx.x >> 1) & 7;
    const size_t t = (size_t)blockIdx.x * RT + (threadIdx.x >> 4);
    const size_t tt = t < n ? t : n - 1;
    FpS px[2], py[2];
    px[0] = load_be48(p0 + sp0 * tt); py[0] = load_be48(p0 + sp0 * tt + 48);
    px[1] = load_be48(p1 + sp1 * tt); py[1] = fp_store(fp_neg(load_be48(p1 + sp1 * tt + 48)));      // -P1
    const P2::RQ q[2] = {{q0 + sq0 * tt, par}, {q1 + sq1 * tt, par}};
    P2::R12 f;
    if (pre) P2::miller_loop2_r<true>(f, px, py, q, pre);
    else P2::miller_loop2_r<false>(f, px, py, q, nullptr);
    if (t < n) row_store12(fbuf, n, t, pr, par, f);
}
// A Verify in two Miller loops (verify_host.inc: verify_sig_side_start): the SIGNATURE side -- e(sig, G2One) for g2pubs, e(G1One, sig) for g1pubs --
// needs nothing of the message, so it runs on a side stream while the message is hashed (k_miller1s_row; strides in bytes, 0 = one broadcast
// record; `pre`: Q is the generator, its prepared lines), and the other pair's loop multiplies its value into the first one's (k_miller1m_row;
// P is negated there, as in k_miller2_row).  The product of the two Miller values is the two-pair loop's value up to nothing: each loop squares its
// own accumulator, (f_a f_b) is what the shared squarings compute.
// [first, end): the tuples of THIS launch (the side stream runs a call of more than 4 096 tuples in pieces of one wave per SIMD, verify_host.inc); n: tuples in fbuf
KERNEL_ROW k_miller1s_row(const u8* p, size_t sp, const u8* q, size_t sq, i32* fbuf, size_t n, const i32* pre, size_t first, size_t end) {
    const int par = threadIdx.x & 1, pr = (threadIdx.x >> 1) & 7;
    const size_t t = first + (size_t)blockIdx.x * RT + (threadIdx.x >> 4);
    const size_t tt = t < end ? t : end - 1;
    const FpS px = load_be48(p + sp * tt), py = load_be48(p + sp * tt + 48);
    P2::R12 f;
    if (pre) P2::miller_loop_table_r(f, px, py, pre);
    else P2::miller_loop_r(f, px, py, P2::RQ{q + sq * tt, par});
    if (t < end) row_store12(fbuf, n, t, pr, par, f);
}
KERNEL_ROW k_miller1m_row(const u8* p, size_t sp, const u8* q, size_t sq, i32* fbuf, size_t n) {
    const int par = threadIdx.x & 1, pr = (threadIdx.x >> 1) & 7;
    const size_t t = (size_t)blockIdx.x * RT + (threadIdx.x >> 4);
    const size_t tt = t < n ? t : n - 1;
    const FpS px = load_be48(p + sp * tt), py = fp_store(fp_neg(load_be48(p + sp * tt + 48)));       // -P
    P2::R12 f;
    P2::miller_loop_r(f, px, py, P2::RQ{q + sq * tt, par});
    f = P2::r12_mul(f, row_load12(fbuf, n, tt, pr, par));
    if (t < n) row_store12(fbuf, n, t, pr, par, f);
}
KERNEL_ROW k_final_exp_is_one_row(const i32* fbuf, const u8* inf_flags, u8* ok, size_t n) {
    const int par = threadIdx.x & 1, pr = (threadIdx.x >> 1) & 7;
    const size_t t = (size_t)blockIdx.x * RT + (threadIdx.x >> 4);
    const size_t tt = t < n ? t : n - 1;
    P2::R12 f = row_load12(fbuf, n, tt, pr, par);
    P2::final_exponentiation_r(f);
    const bool one = P2::r12_is_one(f);
    if (t < n && (threadIdx.x & 15) == 0) ok[t] = (one && !(inf_flags && inf_flags[t])) ? 1 : 0;
}

// unit-level access for the parity tests (blsmi_debug_op with BLSMI_OP_LANE_ROW): records as in k_debug_pairl -- every pair of a row reads
// the whole Fq12, keeps its coefficient, the operation runs in the row layout and pair 0 writes the reassembled result
KERNEL_ROW k_debug_row(int op, const u64* a, const u64* b, u64* out, size_t n) {
    const int par = threadIdx.x & 1;
    const size_t t0 = (size_t)blockIdx.x * RT + (threadIdx.x >> 4);
    const size_t t = t0 < n ? t0 : n - 1;
    P2::Fp12S x, y;
    FpS* cx = reinterpret_cast<FpS*>(&x); FpS* cy = reinterpret_cast<FpS*>(&y);
    for (int j = 0; j < 6; j++) { cx[j] = load_m384(a + (size_t)6 * (12 * t + 2 * j + par)); cy[j] = b ? load_m384(b + (size_t)6 * (12 * t + 2 * j + par)) : cx[j]; }
    if (op >= BLSMI_OP_ROW_DBL_STEP && op <= BLSMI_OP_ROW_ADD_STEP_REF) {     // one Miller-loop step: (X, Y, Z, xq, yq, (xP, yP)) -> (X3, Y3, Z3, c0, c1, c4)
        P2::G2Proj r; r.x = x.c0.c0; r.y = x.c0.c1; r.z = x.c0.c2;
        const P2::Fp2S qx = x.c1.c0, qy = x.c1.c1;
        const FpS px = load_m384(a + (size_t)6 * (12 * t + 10)), py = load_m384(a + (size_t)6 * (12 * t + 11));
        P2::RLine l;
        if (op == BLSMI_OP_ROW_DBL_STEP) P2::r_doubling_step<false>(r, P2::r_point(px, py), l);
        else if (op == BLSMI_OP_ROW_ADD_STEP) P2::r_addition_step<false>(r, P2::RQval{qx, qy}, P2::r_point(px, py), l);
        else {
            P2::Fp2S o0, o1, o2;
            if (op == BLSMI_OP_ROW_DBL_STEP_REF) P2::doubling_step_h(r, o0, o1, o2); else P2::addition_step_h(r, qx, qy, o0, o1, o2);
            l.c0 = o2; l.c1 = P2::fp2_norm(P2::fp2_mul_fp(o1, px)); l.c4 = P2::fp2_norm(P2::fp2_mul_fp(o0, py));
        }
        const FpS res[6] = {r.x.c, r.y.c, r.z.c, l.c0.c, fp_relabel<1, FpS::V>(l.c1.c), fp_relabel<1, FpS::V>(l.c4.c)};
        if (t0 < n && (threadIdx.x & 14) == 0)
            for (int j = 0; j < 6; j++) store_m384(out + (size_t)6 * (12 * t + 2 * j + par), res[j]);
        return;
    }
    if (op >= BLSMI_OP_ROW_G2_DOUBLE && op <= BLSMI_OP_ROW_CLEAR_H2) {       // row_g2.inc: (X1, Y1, Z1, X2, Y2, Z2) -> (X3, Y3, Z3, 0, 0, 0)
        P2::RJ p, q; p.x = x.c0.c0; p.y = x.c0.c1; p.z = x.c0.c2; q.x = x.c1.c0; q.y = x.c1.c1; q.z = x.c1.c2;
        const P2::RJ r = op == BLSMI_OP_ROW_G2_DOUBLE ? P2::r_jdouble(p) : op == BLSMI_OP_ROW_G2_ADD ? P2::r_jadd(p, q) : P2::r_clear_h2(p);
        const FpS zero = fp_zero();
        const FpS res[6] = {r.x.c, r.y.c, r.z.c, zero, zero, zero};
        if (t0 < n && (threadIdx.x & 14) == 0)
            for (int j = 0; j < 6; j++) store_m384(out + (size_t)6 * (12 * t + 2 * j + par), res[j]);
        return;
    }
    const P2::R12 rx = P2::r12_from_pair(x), ry = P2::r12_from_pair(y);
    P2::R12 r;
    switch (op) {
        case BLSMI_OP_FQ12_MUL: r = P2::r12_mul(rx, ry); break;
        case BLSMI_OP_FQ12_SQR: r = P2::r12_sqr(rx); break;
        case BLSMI_OP_FQ12_INV: r = P2::r12_inv(rx); break;
        case BLSMI_OP_FQ12_FROB1: r = P2::r12_frob1(rx); break;
        case BLSMI_OP_FQ12_FROB2: r = P2::r12_frob2(rx); break;
        case BLSMI_OP_FQ12_FROB3: r = P2::r12_frob3(rx); break;
        case BLSMI_OP_FQ12_CYCLO_SQR: r = P2::r12_cyc_sqr(rx); break;
        default: r = P2::r12_mul_by_014(rx, P2::fp2_tight(y.c0.c0), P2::fp2_tight(y.c0.c1), P2::fp2_tight(y.c0.c2)); break;   // BLSMI_OP_FQ12_MUL_BY_014
    }
    const P2::Fp12S z = P2::r12_to_pair(r);
    const FpS* cz = reinterpret_cast<const FpS*>(&z);
    if (t0 < n && (threadIdx.x & 14) == 0)
        for (int j = 0; j < 6; j++) store_m384(out + (size_t)6 * (12 * t + 2 * j + par), cz[j]);
}

// ... and with EIGHT lanes per message, two messages a row (oct_g2.inc): eight messages per 64-lane workgroup
KERNEL_ROW k_clear_h2_oct(const i32* jbuf, u8* good, u8* out, size_t n) {
    hash_prio();
    const int par = threadIdx.x & 1;
    const size_t t = (size_t)blockIdx.x * (WG / 8) + (threadIdx.x >> 3);
    const size_t tt = t < n ? t : n - 1;
    P2::RJ p;
    p.x = P2::wrap(soa_load(jbuf, n, tt, 0 + par)); p.y = P2::wrap(soa_load(jbuf, n, tt, 2 + par)); p.z = P2::wrap(soa_load(jbuf, n, tt, 4 + par));
    const P2::RH r = P2::o_clear_h2_hom(P2::o_jac_to_hom(p));
    P2::Fp2S ax, ay; bool zero;
    P2::o_hto_affine(r, ax, ay, zero);
    if (t < n && (threadIdx.x & 6) == 0) {                                  // pair 0 of the half writes: x.c_par at +48 par, y.c_par at +96 + 48 par
        if (zero) { if (!par) good[t] = 0; }
        else if (good[t]) { u8* o = out + 192 * t; store_be48(o + 48 * par, ax.c); store_be48(o + 96 + 48 * par, ay.c); }
    }
}

// ---- the tail of HashG2 for a few thousand messages (row_g2.inc): k_hash_g2_front (k_hash_pair.hip) leaves the isogeny's Jacobian image of every message in `jbuf`
// (6 Fq per message, structure of arrays like the Miller-loop hand-off; element 2 c + parity of coordinate c) and good[t]; this kernel clears the cofactor
// with sixteen lanes per message and writes the affine wire record.  A message whose result is the point at infinity (Z = 0) gets good[t] = 0 and is
// redone by k_hash_g2_redo, which follows the reference's special cases.
KERNEL_ROW k_clear_h2_row(const i32* jbuf, u8* good, u8* out, size_t n) {
    hash_prio();
    const int par = threadIdx.x & 1;
    const size_t t = (size_t)blockIdx.x * RT + (threadIdx.x >> 4);
    const size_t tt = t < n ? t : n - 1;
    P2::RJ p;
    p.x = P2::wrap(soa_load(jbuf, n, tt, 0 + par)); p.y = P2::wrap(soa_load(jbuf, n, tt, 2 + par)); p.z = P2::wrap(soa_load(jbuf, n, tt, 4 + par));
    const P2::RH r = P2::r_clear_h2_hom(P2::r_jac_to_hom(p));              // homogeneous coordinates on the way: two product times a step (row_g2.inc)
    P2::Fp2S ax, ay; bool zero;
    P2::r_hto_affine(r, ax, ay, zero);
    if (t < n && (threadIdx.x & 14) == 0) {                                 // pair 0 of the row writes: x.c_par at +48 par, y.c_par at +96 + 48 par
        if (zero) { if (!par) good[t] = 0; }
        else if (good[t]) { u8* o = out + 192 * t; store_be48(o + 48 * par, ax.c); store_be48(o + 96 + 48 * par, ay.c); }
    }
}
