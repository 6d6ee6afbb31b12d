// k_pairing_quad.hip -- pairing kernels in the LANE-QUAD layout (quad_body.inc): four adjacent lanes per tuple, 16 tuples per
// 64-lane workgroup.  For batches that do not fill the chip in the lane-pair layout: 16 384 tuples are 1 024 waves here -- one per
// SIMD -- and every lane carries half a lane pair's work.  One wave per SIMD is the design point (larger batches go to the
// lane-pair kernels), so the kernels are compiled for the whole 512-entry register file: the 45-word accumulator, its partner
// half and the operands of an Fq6 product stay in registers.
#ifndef BLSMI_QUAD_WAVES
#define BLSMI_QUAD_WAVES 1
#endif
// measured at 16 384 pairings (same box, tools/ab_bench.py): the Fq12-level routines expanded in line 7.11 -> 6.87 ms (the accumulator
// stays in registers: 512 of them at one wave per SIMD); the 16-squaring run of |x| NOT compressed 4.05 -> 3.95 ms for the final
// exponentiation (a decompression -- one inversion, which does not split over the pairs -- outweighs 16 x (3 - 2) products here);
// assembly-blob cores 7.5 ms and two waves per SIMD (256 registers) 7.6 ms: both off
#ifndef BLSMI_QUAD_NO_INLINE
#define BLSMI_QUAD_INLINE
#endif
#ifndef BLSMI_QUAD_MIN_RUN
#define BLSMI_QUAD_MIN_RUN 27
#endif
#include "pairing.cuh"
#include "device_io.cuh"
namespace blsmi {
namespace pairl {
#include "quad_body.inc"
}  // namespace pairl
}  // namespace blsmi

#define KERNEL_QUAD __global__ void __launch_bounds__(WG, BLSMI_QUAD_WAVES)
namespace P2 = blsmi::pairl;
constexpr int QT = WG / 4;                                               // tuples per workgroup

// the hand-off buffer between Miller loop and final exponentiation is the lane-pair kernels' (device_io.cuh: soa_store): Fq number
// e = 2 (3 half + j) + parity of tuple t.  A lane holds the three Fq2 coefficients j of its half.
BLSMI_DEV void quad_store12(i32* buf, size_t n, size_t t, int half, int par, const P2::Q12& f) {
    const FpS* c = reinterpret_cast<const FpS*>(&f.h);
#pragma unroll
    for (int j = 0; j < 3; j++) soa_store(buf, n, t, 2 * (3 * half + j) + par, c[j]);
}
BLSMI_DEV P2::Q12 quad_load12(const i32* buf, size_t n, size_t t, int half, int par) {
    P2::Q12 f;
    FpS* c = reinterpret_cast<FpS*>(&f.h);
#pragma unroll
    for (int j = 0; j < 3; j++) c[j] = soa_load(buf, n, t, 2 * (3 * half + j) + par);
    return f;
}
// Pairing(P, Q) = FinalExponentiation(MillerLoop) (pairing.go:132-136): the Miller value by the homogeneous steps, as k_miller1h_pair
KERNEL_QUAD k_miller1h_quad(const u8* g1, const u8* g2, i32* fbuf, size_t n) {
    const int par = threadIdx.x & 1, half = (threadIdx.x >> 1) & 1;
    const size_t t = (size_t)blockIdx.x * QT + (threadIdx.x >> 2);
    const size_t tt = t < n ? t : n - 1;                                  // all four lanes of a quad stay active (DPP exchanges)
    const FpS px = load_be48(g1 + 96 * tt), py = load_be48(g1 + 96 * tt + 48);
    const P2::Fp2S qx = P2::wrap(load_be48(g2 + 192 * tt + 48 * par)), qy = P2::wrap(load_be48(g2 + 192 * tt + 96 + 48 * par));
    P2::Q12 f;
    P2::miller_loop_q(f, px, py, qx, qy);
    if (t < n) quad_store12(fbuf, n, t, half, par, f);
}
// pairing.go:79-129 on the hand-off buffer; out = the reference's in-memory FQ12 (72 u64 per tuple).  mode 1: no exponentiation
// (format conversion only, as k_final_exp_pair)
KERNEL_QUAD k_final_exp_quad(const i32* fbuf, u64* out, size_t n, int mode) {
    const int par = threadIdx.x & 1, half = (threadIdx.x >> 1) & 1;
    const size_t t = (size_t)blockIdx.x * QT + (threadIdx.x >> 2);
    const size_t tt = t < n ? t : n - 1;
    P2::Q12 f = quad_load12(fbuf, n, tt, half, par);
    if (mode == 0) P2::final_exponentiation_q(f);
    if (t >= n) return;
    const FpS* c = reinterpret_cast<const FpS*>(&f.h);
#pragma unroll
    for (int j = 0; j < 3; j++) store_m384(out + 72 * t + 6 * (2 * (3 * half + j) + par), c[j]);
}
// CompareTwoPairings in the quad layout (k_miller2_pair's arguments: strides in bytes, 0 = one broadcast record; P1 is negated here;
// `pre`: pair 0's G2 point is the generator, read its prepared lines)
KERNEL_QUAD k_miller2_quad(const u8* p0, size_t sp0, const u8* q0, size_t sq0, const u8* p1, size_t sp1, const u8* q1, size_t sq1, i32* fbuf, size_t n, const i32* pre) {
    const int par = threadIdx.x & 1, half = (threadIdx.x >> 1) & 1;
    const size_t t = (size_t)blockIdx.x * QT + (threadIdx.x >> 2);
    const size_t tt = t < n ? t : n - 1;
    FpS px[2], py[2]; P2::Fp2S qx[2], qy[2];
    px[0] = load_be48(p0 + sp0 * tt); py[0] = load_be48(p0 + sp0 * tt + 48);
    qx[0] = P2::wrap(load_be48(q0 + sq0 * tt + 48 * par)); qy[0] = P2::wrap(load_be48(q0 + sq0 * tt + 96 + 48 * par));
    px[1] = load_be48(p1 + sp1 * tt); py[1] = fp_store(fp_neg(load_be48(p1 + sp1 * tt + 48)));      // -P1
    qx[1] = P2::wrap(load_be48(q1 + sq1 * tt + 48 * par)); qy[1] = P2::wrap(load_be48(q1 + sq1 * tt + 96 + 48 * par));
    P2::Q12 f;
    if (pre) P2::miller_loop2_q<true>(f, px, py, qx, qy, pre);
    else P2::miller_loop2_q<false>(f, px, py, qx, qy, nullptr);
    if (t < n) quad_store12(fbuf, n, t, half, par, f);
}
// The Miller loops of a VerifyAggregate (k_miller1x2_pair in the quad layout): a quad takes TWO consecutive tuples as one 2-pair loop;
// m = ceil(n / 2) values.  The quad that holds the last tuple of an odd n runs a one-pair loop on it.
KERNEL_QUAD k_miller1x2_quad(const u8* g1, const u8* g2, i32* fbuf, size_t n, size_t m) {
    const int par = threadIdx.x & 1, half = (threadIdx.x >> 1) & 1;
    const size_t t = (size_t)blockIdx.x * QT + (threadIdx.x >> 2);
    const size_t tt = t < m ? t : m - 1;
    const size_t i0 = 2 * tt, i1 = 2 * tt + 1 < n ? 2 * tt + 1 : i0;
    FpS px[2], py[2]; P2::Fp2S qx[2], qy[2];
    px[0] = load_be48(g1 + 96 * i0); py[0] = load_be48(g1 + 96 * i0 + 48);
    px[1] = load_be48(g1 + 96 * i1); py[1] = load_be48(g1 + 96 * i1 + 48);
    qx[0] = P2::wrap(load_be48(g2 + 192 * i0 + 48 * par)); qy[0] = P2::wrap(load_be48(g2 + 192 * i0 + 96 + 48 * par));
    qx[1] = P2::wrap(load_be48(g2 + 192 * i1 + 48 * par)); qy[1] = P2::wrap(load_be48(g2 + 192 * i1 + 96 + 48 * par));
    P2::Q12 f;
    P2::miller_loop2_q<false>(f, px, py, qx, qy, nullptr);
    if (!__all(2 * tt + 1 < n)) {                                          // the wave that holds the odd tuple out
        P2::Q12 f1;
        P2::miller_loop_q(f1, px[0], py[0], qx[0], qy[0]);
        const i32 both = 2 * tt + 1 < n ? -1 : 0;
        FpS* a = reinterpret_cast<FpS*>(&f.h); const FpS* b = reinterpret_cast<const FpS*>(&f1.h);
        for (int j = 0; j < 3; j++) a[j] = fp_select(both, a[j], b[j]);
    }
    if (t < m) quad_store12(fbuf, m, t, half, par, f);
}
KERNEL_QUAD k_final_exp_is_one_quad(const i32* fbuf, const u8* inf_flags, u8* ok, size_t n) {
    const int par = threadIdx.x & 1, half = (threadIdx.x >> 1) & 1;
    const size_t t = (size_t)blockIdx.x * QT + (threadIdx.x >> 2);
    const size_t tt = t < n ? t : n - 1;
    P2::Q12 f = quad_load12(fbuf, n, tt, half, par);
    P2::final_exponentiation_q(f);
    const bool one = P2::q12_is_one(f);
    if (t < n && (threadIdx.x & 3) == 0) ok[t] = (one && !(inf_flags && inf_flags[t])) ? 1 : 0;
}

// unit-level access for the parity tests (blsmi_debug_op with BLSMI_OP_LANE_QUAD): records as in k_debug_pairl -- both pairs of a
// quad read the whole Fq12, split it, run the operation in the quad layout and pair A writes the reassembled result
KERNEL_QUAD k_debug_quad(int op, const u64* a, const u64* b, u64* out, size_t n) {
    const int par = threadIdx.x & 1;
    const size_t t0 = (size_t)blockIdx.x * QT + (threadIdx.x >> 2);
    const size_t t = t0 < n ? t0 : n - 1;
    P2::Fp12S x, y;
    FpS* cx = reinterpret_cast<FpS*>(&x); FpS* cy = reinterpret_cast<FpS*>(&y);
    for (int j = 0; j < 6; j++) { cx[j] = load_m384(a + (size_t)6 * (12 * t + 2 * j + par)); cy[j] = b ? load_m384(b + (size_t)6 * (12 * t + 2 * j + par)) : cx[j]; }
    const P2::Q12 qx = P2::q12_from_pair(x), qy = P2::q12_from_pair(y);
    P2::Q12 r;
    switch (op) {
        case BLSMI_OP_FQ12_MUL: P2::q12_mul(r, qx, qy); break;
        case BLSMI_OP_FQ12_SQR: P2::q12_sqr(r, qx); break;
        case BLSMI_OP_FQ12_INV: P2::q12_inv(r, qx); break;
        case BLSMI_OP_FQ12_FROB1: P2::q12_frob1(r, qx); break;
        case BLSMI_OP_FQ12_FROB2: P2::q12_frob2(r, qx); break;
        case BLSMI_OP_FQ12_FROB3: P2::q12_frob3(r, qx); break;
        case BLSMI_OP_FQ12_CYCLO_SQR: P2::q12_cyc_sqr(r, qx); break;
        case BLSMI_OP_FQ12_CYCLO_RUN16: P2::q12_cyc_sqr_run(r, qx, 16); break;
        default: r = P2::q12_mul_by_014(qx, y.c0.c0, y.c0.c1, y.c0.c2); break;          // BLSMI_OP_FQ12_MUL_BY_014
    }
    const P2::Fp12S z = P2::q12_to_pair(r);
    const FpS* cz = reinterpret_cast<const FpS*>(&z);
    if (t0 < n && (threadIdx.x & 2) == 0)
        for (int j = 0; j < 6; j++) store_m384(out + (size_t)6 * (12 * t + 2 * j + par), cz[j]);
}
