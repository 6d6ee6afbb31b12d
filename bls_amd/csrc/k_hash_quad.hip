// k_hash_quad.hip -- the tail of HashG2 in the LANE-QUAD layout (quad_g2.inc): four adjacent lanes per message, 16 messages per 64-lane workgroup.
// k_hash_g2_front (k_hash_pair.hip: the two maps, their sum, the 3-isogeny; a lane pair per message) leaves the isogeny's Jacobian image of every message
// in `jbuf`; this kernel clears the cofactor (hash.go:368-389) and writes the affine wire record.  For 4 097 .. 16 384 messages: 16 384 are 1 024 waves, one on
// every SIMD, each lane carrying 4 / 7 of a doubling's and half an addition's products of the lane-pair kernel (k_hash_g2_pair: 512 waves, half the SIMDs idle).
#include "pairing.cuh"
#include "device_io.cuh"
namespace blsmi {
namespace pairl {
#include "quad_body.inc"
#include "quad_g2.inc"
}  // namespace pairl
#include "quad_g1.inc"
}  // namespace blsmi

namespace P2 = blsmi::pairl;
constexpr int QT = WG / 4;                                               // messages per workgroup

// A message whose result has Z = 0 -- infinity met on the way, an addition of equal x -- gets good[t] = 0 and is redone by k_hash_g2_redo with the
// reference's special cases (as k_clear_h2_row, k_pairing_row.hip)
__global__ void __launch_bounds__(WG, 1) k_clear_h2_quad(const i32* jbuf, u8* good, u8* out, size_t n) {
    hash_prio();
    const int par = threadIdx.x & 1;
    const size_t t = (size_t)blockIdx.x * QT + (threadIdx.x >> 2);
    const size_t tt = t < n ? t : n - 1;
    P2::QJ p;
    p.x = P2::wrap(soa_load(jbuf, n, tt, 0 + par)); p.y = P2::wrap(soa_load(jbuf, n, tt, 2 + par)); p.z = P2::wrap(soa_load(jbuf, n, tt, 4 + par));
    const P2::QJ r = P2::q_clear_h2(p);
    P2::Fp2S ax, ay; bool zero;
    P2::q_jto_affine(r, ax, ay, zero);
    if (t < n && (threadIdx.x & 2) == 0) {                                  // pair A of the quad writes: x.c_par at +48 par, y.c_par at +96 + 48 par
        if (zero) { if (!par) good[t] = 0; }
        else if (good[t]) { u8* o = out + 192 * t; store_be48(o + 48 * par, ax.c); store_be48(o + 96 + 48 * par, ay.c); }
    }
}

// The tail of HashG1 with four lanes per message (quad_g1.inc) behind k_swu_g1_two_lanes (k_hash.hip: pts = the two mapped points of every message in the wire format):
// their sum, the 11-isogeny, the cofactor clearing, the affine record.  1 280 .. 32 768 messages: 0.96 ms in k_hash_g1_finish (a message per lane: 64 .. 256 waves for 4 096 .. 16 384),
// 0.48 ms here.  good[t] = 0: the message met an exception (Z = 0) and is redone by k_hash_g1_finish_redo
__global__ void __launch_bounds__(WG, 2) k_hash_g1_finish_quad(const u8* pts, u8* good, u8* out, size_t n) {
    hash_prio();
    const size_t t = (size_t)blockIdx.x * QT + (threadIdx.x >> 2);
    const size_t tt = t < n ? t : n - 1;
    const Q1Lane ln;
    G1Q p1, p2;
    p1.x = load_be48(pts + 192 * tt); p1.y = load_be48(pts + 192 * tt + 48); p1.z = fp_one();
    p2.x = load_be48(pts + 192 * tt + 96); p2.y = load_be48(pts + 192 * tt + 144); p2.z = fp_one();
    const G1H r = q1h_clear_h(ln, q1_jac_to_hom(ln, q1_iso11(ln, q1_add(ln, p1, p2))));   // the cofactor in homogeneous coordinates: 144 product times instead of 219
    FpS ax, ay; bool zero;
    q1h_to_affine(ln, r, ax, ay, zero);
    if (t < n && (threadIdx.x & 3) == 0) {
        good[t] = zero ? 0 : 1;
        if (!zero) { store_be48(out + 96 * t, ax); store_be48(out + 96 * t + 48, ay); }
    }
}

// unit-level access (blsmi_debug_op, BLSMI_OP_LANE_QUAD | BLSMI_OP_ROW_G2_*): the 12-Fq records of k_debug_row, (X1, Y1, Z1, X2, Y2, Z2) -> (X3, Y3, Z3, 0, 0, 0)
__global__ void __launch_bounds__(WG, 1) k_debug_quad_g2(int op, const u64* a, u64* out, size_t n) {
    const int par = threadIdx.x & 1;
    const size_t t0 = (size_t)blockIdx.x * QT + (threadIdx.x >> 2);
    const size_t t = t0 < n ? t0 : n - 1;
    FpS c[6];
    for (int j = 0; j < 6; j++) c[j] = load_m384(a + (size_t)6 * (12 * t + 2 * j + par));
    P2::QJ p, q; p.x = P2::wrap(c[0]); p.y = P2::wrap(c[1]); p.z = P2::wrap(c[2]); q.x = P2::wrap(c[3]); q.y = P2::wrap(c[4]); q.z = P2::wrap(c[5]);
    const P2::QJ r = op == BLSMI_OP_ROW_G2_DOUBLE ? P2::q_jdouble(p) : op == BLSMI_OP_ROW_G2_ADD ? P2::q_jadd(p, q) : P2::q_clear_h2(p);
    const FpS zero = fp_zero();
    const FpS res[6] = {r.x.c, r.y.c, r.z.c, zero, zero, zero};
    if (t0 < n && (threadIdx.x & 2) == 0)
        for (int j = 0; j < 6; j++) store_m384(out + (size_t)6 * (12 * t + 2 * j + par), res[j]);
}
