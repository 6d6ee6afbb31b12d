// hash.cuh -- the reference's draft-era hash-to-curve on the device, one message per lane.
// Replaces hp/hp2 (hash.go:41-113), optimizedSWUMapHelper (g1.go:628-714), OptimizedSWU2MapHelper
// (g2.go:933-1031), iso11/iso3 (hash.go:185-303), ClearH/clearH2/psi (hash.go:306-389), HashG1/HashG2
// (hash.go:326-331, 405-411) and HashG2WithDomain (g2.go:1041-1085).
// Data-dependent branches of the reference (is g(x0) a square? which sign?) become selects: the
// control flow is the same for all 64 lanes.  The algebra is rearranged wherever the result is the same field
// element or group point ("Hash-to-curve and verify-path algebra"): one exponentiation serves the
// inversion and the square root of an SWU helper, Fq2 roots come from two Fq exponentiations through the norm, the
// isogeny and the cofactor clearing run on Jacobian coordinates (one inversion per hash), ScaleByCofactor goes
// through clearH2 and a psi ladder.  The reference-shaped routines are kept for the inputs those identities exclude.
#pragma once
#include "curve.cuh"

namespace blsmi {

// ---- SHA-256 (the reference uses Go's crypto/sha256, hash.go:4) -------------------------------------
__constant__ const u32 SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

BLSMI_DEV u32 rotr(u32 x, int n) { return __builtin_rotateright32(x, n); }
__device__ __noinline__ void sha256_block(u32 h[8], const u32 win[16]) {
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = win[i];
    u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const u32 s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3), s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const u32 t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + SHA_K[i] + w[i & 15];
        const u32 t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
BLSMI_DEV void sha256_init(u32 h[8]) {
    h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a; h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
}
// digest of (optional 1-byte prefix) || msg[0..len) read bytewise from global memory
__device__ __noinline__ void sha256_msg(u32 h[8], int has_prefix, u8 prefix, const u8* msg, size_t len) {
    sha256_init(h);
    const size_t total = len + (has_prefix ? 1 : 0);
    const size_t padded = ((total + 8) / 64 + 1) * 64;                   // bytes after padding
    u32 w[16];
    for (size_t base = 0; base < padded; base += 64) {
        for (int i = 0; i < 16; i++) {
            u32 word = 0;
            for (int k = 0; k < 4; k++) {
                const size_t pos = base + 4 * i + k;
                u32 byte;
                if (pos < total) byte = has_prefix ? (pos == 0 ? prefix : msg[pos - 1]) : msg[pos];
                else if (pos == total) byte = 0x80;
                else if (pos >= padded - 8) byte = (u32)(((u64)total * 8) >> (8 * (padded - 1 - pos))) & 0xff;
                else byte = 0;
                word = (word << 8) | byte;
            }
            w[i] = word;
        }
        sha256_block(h, w);
    }
}
// digest of the 35-byte string  d[0..8) (32 bytes) || b0 || b1 || b2   (m' || i || j of hp/hp2)
BLSMI_DEV void sha256_35(u32 out[8], const u32 d[8], u32 b0, u32 b1, u32 b2) {
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = d[i];
    w[8] = (b0 << 24) | (b1 << 16) | (b2 << 8) | 0x80;
#pragma unroll
    for (int i = 9; i < 15; i++) w[i] = 0;
    w[15] = 35 * 8;
    sha256_init(out);
    sha256_block(out, w);
}
// 64 big-endian bytes (two digests, t = d1 || d2) mod q as a Montgomery element (hash.go:66-71)
BLSMI_DEV FpS fq_from_two_digests(const u32 d1[8], const u32 d2[8]) {
    u32 hi[12], lo[12];
#pragma unroll
    for (int j = 0; j < 8; j++) { hi[j] = d1[7 - j]; lo[j] = d2[7 - j]; }
#pragma unroll
    for (int j = 8; j < 12; j++) { hi[j] = 0; lo[j] = 0; }
    const FpS h = fp_from_words(hi), l = fp_from_words(lo);
    return fp_store(fp_add(fp_mul(h, C_TWO256), l));
}
// hash.go:41-72 with the digest of (0x01 || msg) computed once (the reference recomputes it per call)
BLSMI_DEV FpS hp_from_digest(const u32 msg_digest[8], u32 ctr) {
    u32 d1[8], d2[8];
    sha256_35(d1, msg_digest, ctr, 1, 1);
    sha256_35(d2, msg_digest, ctr, 1, 2);
    return fq_from_two_digests(d1, d2);
}
// hash.go:74-113
BLSMI_DEV Fp2S hp2_from_digest(const u32 msg_digest[8], u32 ctr) {
    u32 d1[8], d2[8];
    Fp2S r;
    sha256_35(d1, msg_digest, ctr, 1, 1); sha256_35(d2, msg_digest, ctr, 1, 2);
    r.c0 = fq_from_two_digests(d1, d2);
    sha256_35(d1, msg_digest, ctr, 2, 1); sha256_35(d2, msg_digest, ctr, 2, 2);
    r.c1 = fq_from_two_digests(d1, d2);
    return r;
}

// ---- sign helpers (g1.go:621-626, g2.go:916-931): comparisons are on normal-form values ------------
// all-ones iff the normal form of x is > (q-1)/2 ; *is_zero = all-ones iff x == 0
template <int L, int V>
BLSMI_DEV i32 fp_gt_half(const Fp<L, V>& x, i32* is_zero = nullptr) {
    const FpC c = fp_canon(fp_mul(fp_store(x), C_RAW_ONE));             // normal form, canonical limbs
    i32 b = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) { b = (C_QM1O2_RAW.v[i] - c.v[i] + b) >> LB; nz |= c.v[i]; }
    if (is_zero) *is_zero = nz ? 0 : -1;
    return b;                                                              // borrow <=> (q-1)/2 < x
}
BLSMI_DEV FpS fp_sign(const FpS& x) { return fp_select(fp_gt_half(x), C_NEGONE, C_ONE); }         // g1.go:621-626
BLSMI_DEV i32 fp2_sign_is_neg(const Fp2S& f) {                                                     // g2.go:916-931: -1 <=> all-ones
    i32 z1, z0;
    const i32 g1 = fp_gt_half(f.c1, &z1), g0 = fp_gt_half(f.c0, &z0);
    return g1 | (z1 & g0);
}

// ---- G1: simplified SWU to the 11-isogenous curve (g1.go:628-714) ----------------------------------
// The map as the reference writes it: an inversion for x0 and a square root (one exponentiation each).  Kept for the
// rare lanes the fused version below cannot serve (exceptional t, g(x0) = 0).
__device__ __noinline__ void swu_g1_helper_ref(G1Aff& out, const FpS& t) {
    const FpS tsq = fp_store(fp_sqr(t));
    const FpS ndc = fp_store(fp_sub(fp_sqr(tsq), tsq));                   // (-1)^2 t^4 + (-1) t^2
    const i32 ndc0 = fp_is_zero(ndc) ? -1 : 0;
    // x0 = -B (ndc+1) / (A ndc)      or  B / (-A) when ndc == 0 (g1.go:645-659)
    const FpS num = fp_select(ndc0, FpS(C_ELLPB), fp_store(fp_mul(fp_neg(C_ELLPB), fp_add(ndc, C_ONE))));
    const FpS den = fp_select(ndc0, fp_store(fp_neg(C_ELLPA)), fp_store(fp_mul(C_ELLPA, ndc)));
    const FpS x0 = fp_store(fp_mul(num, fp_inv(den)));
    const FpS gx0 = fp_store(fp_add(fp_add(fp_mul(fp_sqr(x0), x0), fp_mul(C_ELLPA, x0)), C_ELLPB));
    bool ok0, ok1;
    const FpS y0 = fp_sqrt(gx0, ok0);
    const FpS x1 = fp_store(fp_mul(fp_neg(tsq), x0));                      // (-1) t^2 x0
    const FpS gx1 = fp_store(fp_add(fp_add(fp_mul(fp_sqr(x1), x1), fp_mul(C_ELLPA, x1)), C_ELLPB));
    // When gx0 is a non-residue, y0 = gx0^((q+1)/4) squares to -gx0, and gx1 = (-1)^3 t^6 gx0 = (t^3 y0)^2: the second
    // root costs three multiplications instead of an exponentiation.  (Either root serves: the sign is fixed below.)
    // The identity needs the regular x0; the exceptional t (ndc == 0: t in {0, 1, -1}) takes the reference's route.
    FpS y1 = fp_store(fp_mul(fp_mul(tsq, t), y0));
    if (__any(ndc0 != 0)) y1 = fp_select(ndc0, fp_sqrt(gx1, ok1), y1);
    const i32 m0 = ok0 ? -1 : 0;
    const FpS x = fp_select(m0, x0, x1);
    const FpS y = fp_select(m0, y0, y1);
    out.x = x;
    out.y = fp_store(fp_mul(y, fp_mul(fp_sign(y), fp_sign(t))));          // g1.go:706-711
    out.inf = 0;
}
// Same map with ONE exponentiation.  x0 = num/den and g(x0) = U/V with U = num^3 + A num den^2 + B den^3, V = den^3.
// With w = U V^3 and e = w^((q-3)/4):  e^2 w = chi(w) = chi(U/V) = +-1, so
//   y0 = U V e        has  y0^2 = chi * U/V   (the square root of g(x0), or of -g(x0) for a non-residue)
//   1/V = chi U V^2 e^2,   x0 = num den^2 / V
// and for a non-residue g(x0), g(x1) = -t^6 g(x0) = (t^3 y0)^2.  Outputs are the same field elements.
// WAVE: the whole wave works on ONE t (every lane the same values); the exponentiation then runs with one limb per lane (fp_row.cuh)
template <int WAVE>
BLSMI_DEV void swu_g1_helper_t(G1Aff& out, const FpS& t) {
    const FpS tsq = fp_store(fp_sqr(t));
    const FpS ndc = fp_store(fp_sub(fp_sqr(tsq), tsq));                   // (-1)^2 t^4 + (-1) t^2
    const i32 ndc0 = fp_is_zero(ndc) ? -1 : 0;
    const FpS num = fp_select(ndc0, FpS(C_ELLPB), fp_store(fp_mul(fp_neg(C_ELLPB), fp_add(ndc, C_ONE))));   // g1.go:645-659
    const FpS den = fp_select(ndc0, fp_store(fp_neg(C_ELLPA)), fp_store(fp_mul(C_ELLPA, ndc)));
    const FpS den2 = fp_store(fp_sqr(den)), V = fp_store(fp_mul(den2, den));
    const FpS nd2 = fp_store(fp_mul(num, den2));                           // num den^2
    const FpS U = fp_store(fp_add(fp_add(fp_mul(fp_sqr(num), num), fp_mul(C_ELLPA, nd2)), fp_mul(C_ELLPB, V)));
    const FpS V2 = fp_store(fp_sqr(V));
    const FpS UV = fp_store(fp_mul(U, V));
    FpS e;
    if constexpr (WAVE != 0) e = fp_pow_spread<WAVE>(fp_mul(UV, V2), C_QM3O4, BLSMI_QM3O4_BITS);
    else e = fp_pow_qm3o4(fp_mul(UV, V2));
    const FpS y0 = fp_store(fp_mul(UV, e));
    const i32 m0 = fp_eq(fp_mul(fp_sqr(y0), V), U) ? -1 : 0;              // g(x0) is a square
    const FpS vinv = fp_store(fp_mul(fp_mul(UV, V), fp_sqr(e)));           // chi / V
    const FpS x0p = fp_store(fp_mul(nd2, vinv));                           // chi * x0
    const FpS x0 = fp_select(m0, x0p, fp_store(fp_neg(x0p)));
    const FpS x1 = fp_store(fp_mul(fp_neg(tsq), x0));                      // (-1) t^2 x0
    const FpS y1 = fp_store(fp_mul(fp_mul(tsq, t), y0));
    const FpS x = fp_select(m0, x0, x1);
    const FpS y = fp_select(m0, y0, y1);
    out.x = x;
    out.y = fp_store(fp_mul(y, fp_mul(fp_sign(y), fp_sign(t))));          // g1.go:706-711
    out.inf = 0;
    const i32 special = ndc0 | (fp_is_zero(U) ? -1 : 0);
    if (__any(special != 0)) {
        G1Aff r; swu_g1_helper_ref(r, t);
        out.x = fp_select(special, r.x, out.x); out.y = fp_select(special, r.y, out.y);
    }
}
__device__ __noinline__ void swu_g1_helper(G1Aff& out, const FpS& t) { swu_g1_helper_t<0>(out, t); }
__device__ __noinline__ void swu_g1_helper_wave(G1Aff& out, const FpS& t) { swu_g1_helper_t<1>(out, t); }
__device__ __noinline__ void swu_g1_helper_row16(G1Aff& out, const FpS& t) { swu_g1_helper_t<2>(out, t); }   // t uniform over each row of sixteen lanes
template <int N>
BLSMI_DEV FpS horner_fp(const FpS (&c)[N], const FpS& x) {
    FpS v = c[N - 1];
    for (int i = N - 2; i >= 0; i--) v = fp_store(fp_add(fp_mul(v, x), c[i]));
    return v;
}
// The isogeny evaluated on a Jacobian point without inversions.  With x = X/W, W = Z^2, every polynomial is
// homogenised (P_h = sum c_i X^i W^(d-i) = p(x) W^d); deg xnum = deg xden + 1 and deg ynum = deg yden, so
//   x' = XN / (W XD),  y' = Y YN / (Z^3 YD)   and   (X', Y', Z') = (XN XD YD^2,  Y YN XD^3 YD^2,  Z XD YD).
// Same point as iso11/iso3 of the affine image (hash.go:185-206, 282-303), minus two inversions per hash.
template <class F, int NXN, int NXD, int NYN, int NYD>
__device__ __noinline__ void iso_jac(Jac<F>& out, const Jac<F>& p, const F (&xn)[NXN], const F (&xd)[NXD], const F (&yn)[NYN], const F (&yd)[NYD]) {
    static_assert(NXD == NXN - 1 && NYD == NYN && NXN <= NYN, "degree pattern of the 11- and 3-isogeny maps");
    constexpr int D = NYN - 1;
    F wp[D + 1];
    wp[1] = f_store(f_sqr(p.z));
    for (int i = 2; i <= D; i++) wp[i] = f_store(f_mul(wp[i - 1], wp[1]));
    auto hom = [&](const F* c, int d) {
        F v = c[d];
        for (int i = d - 1; i >= 0; i--) v = f_store(f_add(f_mul(v, p.x), f_mul(c[i], wp[d - i])));
        return v;
    };
    const F XN = hom(xn, NXN - 1), XD = hom(xd, NXD - 1), YN = hom(yn, NYN - 1), YD = hom(yd, NYD - 1);
    const F xdyd = f_store(f_mul(XD, YD));
    const F yd2 = f_store(f_sqr(YD)), xd2 = f_store(f_sqr(XD));
    out.z = f_store(f_mul(p.z, xdyd));
    out.x = f_store(f_mul(f_mul(XN, XD), yd2));
    out.y = f_store(f_mul(f_mul(f_mul(p.y, YN), f_mul(xd2, XD)), yd2));
    out.inf = p.inf | (f_is_zero(out.z) ? -1 : 0);
}
// hash.go:185-206
__device__ __noinline__ void iso11(G1Aff& out, const G1Aff& p) {
    const FpS xn = horner_fp(C_XNUM11, p.x), xd = horner_fp(C_XDEN11, p.x), yn = horner_fp(C_YNUM11, p.x), yd = horner_fp(C_YDEN11, p.x);
    const FpS inv = fp_inv(fp_mul(xd, yd));                                // 1/(xd*yd): xn/xd = xn*yd*inv, yn/yd = yn*xd*inv
    out.x = fp_store(fp_mul(fp_mul(xn, yd), inv));
    out.y = fp_store(fp_mul(fp_mul(fp_mul(p.y, yn), xd), inv));
    out.inf = 0;
}
// hash.go:306-321: add the two mapped points, apply the isogeny, clear the cofactor by (|x| + 1)
// The sum stays in Jacobian coordinates through the isogeny and the cofactor multiplication: one inversion (the final
// ToAffine) instead of three.  A sum at infinity (p2 = -p1) follows the reference's affine steps literally.
// clear == 0: WITHOUT the cofactor clearing of hash.go:306-309 -- the point S on E(Fq) whose multiple [1 - x] S is HashG1.  A large g2pubs
// VerifyAggregate pairs S_i instead of H_i and raises the PRODUCT of its Miller values to 1 - x once (verify_host.inc: the reduced pairing is
// bilinear in its E(Fq) argument modulo r E(Fq), so FE(ML(S, Q))^(1-x) = FE(ML([1-x] S, Q))): 63 doublings + 6 additions less per message.
// special (may be null; meaningful with clear == 0): raised (bit 1) when a message's two mapped points cancel (p2 = -p1).  The reference then
// clears the cofactor of a value its affine steps produce out of (0, 0) -- not a curve point, so the identity FE(ML(S, Q))^(1-x) = FE(ML([1-x] S, Q))
// does not cover it: the caller redoes its aggregate with the cleared hash points (verify_host.inc; probability ~2^-381 per message).
__device__ __noinline__ void swu_finish_g1(G1Aff& out, const G1Aff& p1, const G1Aff& p2, int clear = 1, int* special = nullptr) {
    const G1Jac sj = jac_add_affine(to_jac(p1), p2);
    G1Jac ij; iso_jac(ij, sj, C_XNUM11, C_XDEN11, C_YNUM11, C_YDEN11);
    if (clear) out = jac_to_affine(jac_add(jac_mul_u64_public(ij, BLSMI_X_ABS), ij));
    else out = jac_to_affine(ij);
    if (__any(sj.inf != 0)) {
        G1Aff s = jac_to_affine(sj), q, r;
        iso11(q, s);
        r = clear ? jac_to_affine(jac_add_affine(aff_mul_u64_public(q, BLSMI_X_ABS), q)) : q;
        out.x = fp_select(sj.inf, r.x, out.x); out.y = fp_select(sj.inf, r.y, out.y); out.inf = (sj.inf & r.inf) | (~sj.inf & out.inf);
        if (!clear && special && sj.inf) atomicOr(special, 2);
    }
}
__device__ __noinline__ void swu_map_g1(G1Aff& out, const FpS& t1, const FpS& t2, int clear = 1, int* special = nullptr) {
    G1Aff p1, p2;
    swu_g1_helper(p1, t1);
    swu_g1_helper(p2, t2);
    swu_finish_g1(out, p1, p2, clear, special);
}
// hash.go:326-331
__device__ __noinline__ void hash_g1(G1Aff& out, const u8* msg, size_t len, int clear = 1, int* special = nullptr) {
    u32 d[8];
    sha256_msg(d, 1, 0x01, msg, len);
    const FpS t1 = hp_from_digest(d, 0), t2 = hp_from_digest(d, 1);
    swu_map_g1(out, t1, t2, clear, special);
}
// ---- G2 (g2.go:933-1031, hash.go:282-411) ---------------------------------------------------------------
// Reference-shaped version (inversion + norm root + root: three exponentiations); serves g(x0) = 0.
__device__ __noinline__ void swu_g2_helper_ref(G2Aff& out, const Fp2S& t) {
    Fp2S nqr; nqr.c0 = C_ONE; nqr.c1 = C_ONE;
    const Fp2S tsq = fp2_store(fp2_sqr(t));
    const Fp2S nqr_tsq = fp2_store(fp2_mul_nr(tsq));
    const Fp2S ndc = fp2_store(fp2_add(fp2_sqr(nqr_tsq), nqr_tsq));       // nqr^2 t^4 + nqr t^2
    const i32 ndc0 = fp2_is_zero(ndc) ? -1 : 0;
    const Fp2S num = fp2_select(ndc0, Fp2S(C_ELL2PB), fp2_store(fp2_mul(fp2_neg(C_ELL2PB), fp2_add(ndc, fp2_one()))));
    const Fp2S den = fp2_select(ndc0, fp2_store(fp2_mul_nr(C_ELL2PA)), fp2_store(fp2_mul(C_ELL2PA, ndc)));
    const Fp2S x0 = fp2_store(fp2_mul(num, fp2_store(fp2_inv(den))));
    const Fp2S gx0 = fp2_store(fp2_add(fp2_add(fp2_mul(fp2_sqr(x0), x0), fp2_mul(C_ELL2PA, x0)), C_ELL2PB));
    // The reference takes sqrt(gx0) and, when gx0 is not a square, sqrt(gx1) with gx1 = nqr^3 t^6 gx0 (g2.go:977-1019).
    // Exactly one root is needed, so: one Fq exponentiation decides gx0 through its norm n0 (s0 = n0^((q+1)/4),
    // s0^2 = +-n0); for a non-residue, N(gx1) = 8 N(t)^6 n0 = (sqrt(-8) N(t)^3 s0)^2 gives the norm root of gx1 for
    // free; a second exponentiation then finishes the root of whichever was selected (fp2_sqrt_from_norm_root).
    bool ok0;
    FpS n0;
    const FpS s0 = fp2_norm_root(gx0, n0, ok0);
    const Fp2S x1 = fp2_store(fp2_mul(nqr_tsq, x0));
    const Fp2S t6 = fp2_store(fp2_sqr(fp2_mul(tsq, t)));
    Fp2S nqr3 = fp2_store(fp2_mul_nr(fp2_mul_nr(nqr)));                    // nqr^3
    const Fp2S gx1 = fp2_store(fp2_mul(fp2_mul(nqr3, t6), gx0));
    const FpS nt = fp_store(fp_add(fp_sqr(t.c0), fp_sqr(t.c1)));
    const FpS s1 = fp_store(fp_mul(fp_mul(fp_mul(fp_sqr(nt), nt), s0), C_SQRT_M8));
    const i32 m0 = ok0 ? -1 : 0;
    const Fp2S x = fp2_select(m0, x0, x1);
    const Fp2S g = fp2_select(m0, gx0, gx1);
    Fp2S y = fp2_sqrt_from_norm_root(g, fp_select(m0, s0, s1));
    const i32 flip = fp2_sign_is_neg(t) ^ fp2_sign_is_neg(y);             // signT != signY (g2.go:983-988, 1021-1026)
    y = fp2_select(flip, fp2_store(fp2_neg(y)), y);
    out.x = x; out.y = y; out.inf = 0;
}
// Same map with TWO exponentiations: g(x0) = U/V as above (over Fq2); its norm is a/b with a = N(U), b = N(V).
// With w = a b^3, e = w^((q-3)/4):  s0 = a b e has s0^2 = chi * a/b (the norm root fp2_sqrt_from_norm_root needs) and
// 1/b = chi a b^2 e^2, which yields both x0 = num conj(den) N(den)^2 / b and g(x0) = U conj(V) / b without an inversion.
template <int WAVE>
BLSMI_DEV void swu_g2_helper_t(G2Aff& out, const Fp2S& t) {
    Fp2S nqr; nqr.c0 = C_ONE; nqr.c1 = C_ONE;
    const Fp2S tsq = fp2_store(fp2_sqr(t));
    const Fp2S nqr_tsq = fp2_store(fp2_mul_nr(tsq));
    const Fp2S ndc = fp2_store(fp2_add(fp2_sqr(nqr_tsq), nqr_tsq));       // nqr^2 t^4 + nqr t^2
    const i32 ndc0 = fp2_is_zero(ndc) ? -1 : 0;
    const Fp2S num = fp2_select(ndc0, Fp2S(C_ELL2PB), fp2_store(fp2_mul(fp2_neg(C_ELL2PB), fp2_add(ndc, fp2_one()))));
    const Fp2S den = fp2_select(ndc0, fp2_store(fp2_mul_nr(C_ELL2PA)), fp2_store(fp2_mul(C_ELL2PA, ndc)));
    const Fp2S den2 = fp2_store(fp2_sqr(den)), V = fp2_store(fp2_mul(den2, den));
    const Fp2S nd2 = fp2_store(fp2_mul(num, den2));
    const Fp2S U = fp2_store(fp2_add(fp2_add(fp2_mul(fp2_sqr(num), num), fp2_mul(C_ELL2PA, nd2)), fp2_mul(C_ELL2PB, V)));
    const FpS nden = fp_store(fp_add(fp_sqr(den.c0), fp_sqr(den.c1)));    // N(den)
    const FpS nden2 = fp_store(fp_sqr(nden));
    const FpS bb = fp_store(fp_mul(nden2, nden));                           // b = N(V) = N(den)^3
    const FpS aa = fp_store(fp_add(fp_sqr(U.c0), fp_sqr(U.c1)));           // a = N(U)
    const FpS ab = fp_store(fp_mul(aa, bb)), b2 = fp_store(fp_sqr(bb));
    FpS e;
    if constexpr (WAVE != 0) e = fp_pow_spread<WAVE>(fp_mul(ab, b2), C_QM3O4, BLSMI_QM3O4_BITS);
    else e = fp_pow_qm3o4(fp_mul(ab, b2));
    const FpS s0 = fp_store(fp_mul(ab, e));
    const i32 m0 = fp_eq(fp_mul(fp_sqr(s0), bb), aa) ? -1 : 0;             // N(g(x0)) is a square <=> g(x0) is a square
    const FpS binv_p = fp_store(fp_mul(fp_mul(ab, bb), fp_sqr(e)));        // chi / b
    const FpS binv = fp_select(m0, binv_p, fp_store(fp_neg(binv_p)));
    const Fp2S x0 = fp2_store(fp2_mul_fp(fp2_mul(num, fp2_conj(den)), fp_store(fp_mul(nden2, binv))));
    const Fp2S gx0 = fp2_store(fp2_mul_fp(fp2_mul(U, fp2_conj(V)), binv));
    const Fp2S x1 = fp2_store(fp2_mul(nqr_tsq, x0));
    const Fp2S t6 = fp2_store(fp2_sqr(fp2_mul(tsq, t)));
    Fp2S nqr3 = fp2_store(fp2_mul_nr(fp2_mul_nr(nqr)));                    // nqr^3
    const Fp2S gx1 = fp2_store(fp2_mul(fp2_mul(nqr3, t6), gx0));          // g2.go:1005-1010
    const FpS nt = fp_store(fp_add(fp_sqr(t.c0), fp_sqr(t.c1)));
    const FpS s1 = fp_store(fp_mul(fp_mul(fp_mul(fp_sqr(nt), nt), s0), C_SQRT_M8));   // N(gx1) = 8 N(t)^6 N(gx0) = s1^2
    const Fp2S x = fp2_select(m0, x0, x1);
    const Fp2S g = fp2_select(m0, gx0, gx1);
    Fp2S y = fp2_sqrt_from_norm_root<WAVE>(g, fp_select(m0, s0, s1));
    const i32 flip = fp2_sign_is_neg(t) ^ fp2_sign_is_neg(y);             // signT != signY (g2.go:983-988, 1021-1026)
    y = fp2_select(flip, fp2_store(fp2_neg(y)), y);
    out.x = x; out.y = y; out.inf = 0;
    const i32 special = fp_is_zero(aa) ? -1 : 0;
    if (__any(special != 0)) {
        G2Aff r; swu_g2_helper_ref(r, t);
        out.x = fp2_select(special, r.x, out.x); out.y = fp2_select(special, r.y, out.y);
    }
}
__device__ __noinline__ void swu_g2_helper(G2Aff& out, const Fp2S& t) { swu_g2_helper_t<0>(out, t); }
__device__ __noinline__ void swu_g2_helper_wave(G2Aff& out, const Fp2S& t) { swu_g2_helper_t<1>(out, t); }
__device__ __noinline__ void swu_g2_helper_row16(G2Aff& out, const Fp2S& t) { swu_g2_helper_t<2>(out, t); }
template <int N>
BLSMI_DEV Fp2S horner_fp2(const Fp2S (&c)[N], const Fp2S& x) {
    Fp2S v = c[N - 1];
    for (int i = N - 2; i >= 0; i--) v = fp2_store(fp2_add(fp2_mul(v, x), c[i]));
    return v;
}
// hash.go:282-303
__device__ __noinline__ void iso3(G2Aff& out, const G2Aff& p) {
    const Fp2S xn = horner_fp2(C_XNUM3, p.x), xd = horner_fp2(C_XDEN3, p.x), yn = horner_fp2(C_YNUM3, p.x), yd = horner_fp2(C_YDEN3, p.x);
    const Fp2S inv = fp2_store(fp2_inv(fp2_store(fp2_mul(xd, yd))));
    out.x = fp2_store(fp2_mul(fp2_mul(xn, yd), inv));
    out.y = fp2_store(fp2_mul(fp2_mul(fp2_mul(p.y, yn), xd), inv));
    out.inf = 0;
}
// hash.go:341-366
__device__ __noinline__ void psi(G2Aff& out, const G2Aff& g) {
    Fp2S qix = fp2_store(fp2_mul(C_IWSC, g.x));
    qix.c0 = fp_store(fp_mul(qix.c0, C_KQIX));
    qix.c1 = fp_store(fp_neg(fp_mul(qix.c1, C_KQIX)));
    const Fp2S qiy = fp2_store(fp2_mul(C_IWSC, g.y));
    Fp2S q2;
    q2.c0 = fp_store(fp_mul(fp_add(qiy.c0, qiy.c1), C_KQIY));
    q2.c1 = fp_store(fp_mul(fp_sub(qiy.c0, qiy.c1), C_KQIY));
    out.x = fp2_store(fp2_mul_nr(qix));
    out.y = fp2_store(fp2_mul_nr(q2));
    out.inf = g.inf;
}
// hash.go:368-389
__device__ __noinline__ void clear_h2(G2Aff& out, const G2Aff& p) {
    G2Jac work = aff_mul_u64_public(p, BLSMI_X_ABS);
    work = jac_add_affine(work, p);
    G2Aff mpsi; psi(mpsi, p); mpsi = aff_neg(mpsi);
    work = jac_add_affine(work, mpsi);
    work = jac_mul_u64_public(work, BLSMI_X_ABS);
    work = jac_add_affine(work, mpsi);
    work = jac_add_affine(work, aff_neg(p));
    G2Aff p2 = jac_to_affine(jac_double(to_jac(p)));
    G2Aff pp; psi(pp, p2); psi(p2, pp);
    work = jac_add_affine(work, p2);
    out = jac_to_affine(work);
}
// psi on Jacobian coordinates: (X, Y, Z) -> (Cx conj(X), Cy conj(Y), conj(Z)) with Cx = (1+u) kQiX conj(iwsc),
// Cy = (1+u)^2 kQiY conj(iwsc) -- the affine psi above is (x, y) -> (Cx conj(x), Cy conj(y)).
BLSMI_DEV G2Jac psi_jac(const G2Jac& g) {
    G2Jac r;
    r.x = fp2_store(fp2_mul(C_PSI_CX, fp2_conj(g.x)));
    r.y = fp2_store(fp2_mul(C_PSI_CY, fp2_conj(g.y)));
    r.z = fp2_store(fp2_conj(g.z));
    r.inf = g.inf;
    return r;
}
// clearH2 (hash.go:368-389) on a Jacobian input: the same chain with general additions, no intermediate ToAffine
BLSMI_DEV G2Jac clear_h2_jac_j(const G2Jac& p) {
    G2Jac work = jac_mul_u64_public(p, BLSMI_X_ABS);
    work = jac_add(work, p);
    const G2Jac mpsi = jac_neg(psi_jac(p));
    work = jac_add(work, mpsi);
    work = jac_mul_u64_public(work, BLSMI_X_ABS);
    work = jac_add(work, mpsi);
    work = jac_add(work, jac_neg(p));
    return jac_add(work, psi_jac(psi_jac(jac_double(p))));
}
__device__ __noinline__ void clear_h2_jac(G2Aff& out, const G2Jac& p) { out = jac_to_affine(clear_h2_jac_j(p)); }
// ScaleByCofactor (g2.go:104-115, 130-138: a bit-serial multiplication by the 507-bit cofactor h2).  Same point, shorter road:
// clearH2(P) = [3 (x^2 - 1) h2] P on all of E'(Fq2) and lands in G2, where psi acts as x, so [h2] P = [c] Q with Q = clearH2(P) and
// c = (3 (x^2 - 1))^-1 mod r = sum d_i |x|^i, i.e. sum d_i Q_i with Q_i = (-1)^i psi^i(Q).  The four digits are one:
// d_0 = (|x| + 1) / 3, d_1 = 2 d_0 - 1, d_2 = 2 d_0 - 2, d_3 = d_0 - 1 (gen_consts.py asserts it), hence
//     [h2] P = [d_0] (Q_0 + 2 Q_1 + 2 Q_2 + Q_3) - (Q_1 + 2 Q_2 + Q_3):
// ONE 64-bit multiplication (fixed 4-bit windows over five multiples: 63 doublings, 16 additions, curve.cuh) and seven additions after clearH2's two, nothing leaves Jacobian
// coordinates before the end (one inversion).  [Until round 3 the four digits ran as a joint ladder: 64 doublings + 110 mixed additions, and clearH2 ended in an inversion of its own.]
__device__ __noinline__ void scale_by_cofactor_g2(G2Aff& out, const G2Aff& pt) {
    const G2Jac q0 = clear_h2_jac_j(to_jac(pt));
    const G2Jac p1 = psi_jac(q0), q2 = psi_jac(p1), q1 = jac_neg(p1), q3 = jac_neg(psi_jac(q2));
    const G2Jac a = jac_add(q1, q2);
    const G2Jac t = jac_add(jac_add(q0, q3), jac_double(a));
    const G2Jac s = jac_add(jac_add(a, q2), q3);
    out = jac_to_affine(jac_add(jac_mul_h2_d0(t), jac_neg(s)));
}
__device__ __noinline__ void swu_finish_g2(G2Aff& out, G2Aff p1, const G2Aff& p2);
// hash.go:391-411
__device__ __noinline__ void hash_g2(G2Aff& out, const u8* msg, size_t len) {
    u32 d[8];
    sha256_msg(d, 1, 0x01, msg, len);
    const Fp2S t1 = hp2_from_digest(d, 0), t2 = hp2_from_digest(d, 1);
    G2Aff p1, p2;
    swu_g2_helper(p1, t1);
    swu_g2_helper(p2, t2);
    swu_finish_g2(out, p1, p2);
}
__device__ __noinline__ void swu_finish_g2(G2Aff& out, G2Aff p1, const G2Aff& p2) {
    G2Aff s;
    const G2Jac sj = jac_add_affine(to_jac(p1), p2);                      // stays Jacobian through iso3 and clearH2
    G2Jac ij; iso_jac(ij, sj, C_XNUM3, C_XDEN3, C_YNUM3, C_YDEN3);
    clear_h2_jac(out, ij);
    if (__any(sj.inf != 0)) {                                              // p2 = -p1: the reference's affine steps, literally
        G2Aff r;
        s = jac_to_affine(sj);
        iso3(p1, s);
        clear_h2(r, p1);
        out.x = fp2_select(sj.inf, r.x, out.x); out.y = fp2_select(sj.inf, r.y, out.y); out.inf = (sj.inf & r.inf) | (~sj.inf & out.inf);
    }
}

// g2.go:1041-1085: try-and-increment on x0 = (H(m||d||01), H(m||d||02)); favour the y with Parity(),
// then scale by the 507-bit G2 cofactor (g2.go:130-138).  The loop runs until every lane of the wave
// has found a square (uniform trip count; finished lanes keep their result).
// x0 = (H(m || domain || 01), H(m || domain || 02)) as an element of Fq2 (g2.go:1041-1048)
__device__ __noinline__ void tai_g2_x0(Fp2S& x0, const u8* msg32, const u8* domain8) {
    u32 w[16], dre[8], dim[8];
    for (int tag = 1; tag <= 2; tag++) {                                   // SHA-256 of the 41-byte string m || domain || tag
        for (int i = 0; i < 8; i++) w[i] = ((u32)msg32[4 * i] << 24) | ((u32)msg32[4 * i + 1] << 16) | ((u32)msg32[4 * i + 2] << 8) | msg32[4 * i + 3];
        for (int i = 0; i < 2; i++) w[8 + i] = ((u32)domain8[4 * i] << 24) | ((u32)domain8[4 * i + 1] << 16) | ((u32)domain8[4 * i + 2] << 8) | domain8[4 * i + 3];
        w[10] = ((u32)tag << 24) | 0x800000;
        w[11] = w[12] = w[13] = w[14] = 0;
        w[15] = 41 * 8;
        u32* d = tag == 1 ? dre : dim;
        sha256_init(d);
        sha256_block(d, w);
    }
    u32 wre[12], wim[12];
    for (int j = 0; j < 8; j++) { wre[j] = dre[7 - j]; wim[j] = dim[7 - j]; }
    for (int j = 8; j < 12; j++) { wre[j] = 0; wim[j] = 0; }
    x0.c0 = fp_from_words(wre); x0.c1 = fp_from_words(wim);
}
__device__ __noinline__ void hash_g2_with_domain(G2Aff& out, const u8* msg32, const u8* domain8) {
    Fp2S x0; tai_g2_x0(x0, msg32, domain8);
    // The loop only has to DECIDE whether x0^3 + b is a square -- one Fq exponentiation on its norm; the root itself
    // (a second exponentiation) is taken once, after the loop, for the x0 each lane settled on.  The wave iterates until
    // its slowest lane is done (about log2(64) + 1 rounds), so halving the cost of a round matters.
    G2Aff pt; pt.x = x0; pt.y = fp2_one(); pt.inf = 0;
    Fp2S gsel = fp2_zero(); FpS ssel = fp_zero();
    i32 done = 0;
    while (true) {
        const Fp2S gx = fp2_store(fp2_add(fp2_mul(fp2_sqr(x0), x0), C_B2));
        bool ok; FpS nrm;
        const FpS s = fp2_norm_root(gx, nrm, ok);
        const i32 take = (ok ? -1 : 0) & ~done;
        pt.x = fp2_select(take, x0, pt.x);
        gsel = fp2_select(take, gx, gsel);
        ssel = fp_select(take, s, ssel);
        done |= take;
        if (__all(done != 0)) break;
        x0 = fp2_store(fp2_add(x0, fp2_one()));
    }
    {
        Fp2S y = fp2_sqrt_from_norm_root(gsel, ssel);                      // either root: the choice follows
        // favour y with Parity() == true (g2.go:1074-1077): parity(y) <=> y > -y (fq2.go:256-260)
        const i32 y_gt = fp2_sign_is_neg(y);                               // y > (q-1)/2 lexicographically (c1 first) <=> y > -y
        pt.y = fp2_select(y_gt, y, fp2_store(fp2_neg(y)));
    }
    scale_by_cofactor_g2(out, pt);
}

// HashG2WithDomain for large batches: the 64 lanes of a wave SHARE the try-and-increment search of their 64 messages.  In the
// loop above every lane walks its own counter and the wave waits for its unluckiest message (about seven rounds of one Fq
// exponentiation each); here a round hands the wave's lanes to the messages that are still open -- round 0: candidate 0 of every
// message; then 64 / (open messages) consecutive candidates of each open message -- so about half, then a quarter ... of the
// messages close per round and three rounds nearly always do.  The smallest passing counter wins (atomicMin within a round, rounds
// in increasing order): the x the reference's loop stops at.  A helper lane recomputes the message's x0 (two SHA-256 blocks)
// and the winner leaves the norm root for the owner in LDS.  lds: TAI_WAVE_LDS_WORDS words per wave; the caller's workgroup is one wave.
constexpr int TAI_WAVE_LDS_WORDS = 3 * 64 + 64 * NL;
BLSMI_DEV FpS fp_small_mont(u32 c) {                                       // c * 1 in Montgomery form
    FpS acc = fp_zero(), pw = C_ONE;
    for (int i = 0; i < 32; i++) {
        acc = fp_select(-(i32)((c >> i) & 1), fp_store(fp_add(acc, pw)), acc);
        pw = fp_store(fp_add(pw, pw));
    }
    return acc;
}
// tai_g2_wave: the search and the root -- the affine point before ScaleByCofactor; hash_g2_with_domain_wave: the whole hash
__device__ __noinline__ void tai_g2_wave(G2Aff& pt, const u8* msgs32, size_t first, size_t n, const u8* domain8, u32* lds) {
    u32* win = lds; u32* nextc = lds + 64; u32* list = lds + 128; i32* sbuf = reinterpret_cast<i32*>(lds + 192);
    const int lane = (int)(threadIdx.x & 63);
    const bool valid = first + lane < n;
    const size_t own = valid ? first + lane : n - 1;
    Fp2S x0; tai_g2_x0(x0, msgs32 + 32 * own, domain8);
    win[lane] = 0xffffffffu; nextc[lane] = 0;
    bool open = valid;
    __syncthreads();
    while (true) {
        const unsigned long long U = __ballot(open ? 1 : 0);
        if (U == 0) break;
        const int cnt = __popcll(U), k = 64 / cnt;
        if (open) list[__popcll(U & ((1ull << lane) - 1))] = (u32)lane;
        __syncthreads();
        const int slot = lane / k;
        const bool active = slot < cnt;
        const int m = (int)list[active ? slot : 0];                        // the wave-local message this lane works for
        const u32 cand = nextc[m] + (u32)(lane % k);
        Fp2S xm = x0;
        if (U != ~0ull) tai_g2_x0(xm, msgs32 + 32 * (first + m), domain8);  // (all 64 open -- round 0 of a full wave: every lane has its own message)
        xm.c0 = fp_store(fp_add(xm.c0, fp_small_mont(cand)));
        const Fp2S gx = fp2_store(fp2_add(fp2_mul(fp2_sqr(xm), xm), C_B2));
        bool ok; FpS nrm;
        const FpS s = fp2_norm_root(gx, nrm, ok);
        if (active && ok) atomicMin(&win[m], cand);
        __syncthreads();
        if (active && ok && win[m] == cand)
            for (int i = 0; i < NL; i++) sbuf[m * NL + i] = s.v[i];
        if (open) { if (win[lane] != 0xffffffffu) open = false; else nextc[lane] += (u32)k; }
        __syncthreads();
    }
    // every owner: the x its search stopped at, g(x), the norm root its winner left
    pt.x = x0; pt.x.c0 = fp_store(fp_add(x0.c0, fp_small_mont(valid ? win[lane] : 0u))); pt.y = fp2_one(); pt.inf = 0;
    const Fp2S gsel = fp2_store(fp2_add(fp2_mul(fp2_sqr(pt.x), pt.x), C_B2));
    FpS ssel;
    for (int i = 0; i < NL; i++) ssel.v[i] = valid ? sbuf[lane * NL + i] : 0;
    Fp2S y = fp2_sqrt_from_norm_root(gsel, ssel);                          // either root: the choice follows
    const i32 y_gt = fp2_sign_is_neg(y);                                   // favour y with Parity() == true (g2.go:1074-1077)
    pt.y = fp2_select(y_gt, y, fp2_store(fp2_neg(y)));
}
__device__ __noinline__ void hash_g2_with_domain_wave(G2Aff& out, const u8* msgs32, size_t first, size_t n, const u8* domain8, u32* lds) {
    G2Aff pt;
    tai_g2_wave(pt, msgs32, first, n, domain8, lds);
    scale_by_cofactor_g2(out, pt);
}

// The try-and-increment search of HashG2WithDomain (g2.go:1049-1077) with EIGHT lanes per message: lane g of an aligned group of eight tests the candidate x0 + 8 r + g in round r,
// the group takes the smallest passing counter of the first round that has one (the x the reference's loop stops at) -- one round
// almost always (a candidate passes with probability 1/2), where two lanes need 1.33 on average and a wave of 32 messages waits for
// its unluckiest (~3.5 rounds).  Lane 0 of the group returns the point; the others return copies.
BLSMI_DEV FpS fp_from_lane(const FpS& a, int src) { FpS r; for (int i = 0; i < NL; i++) r.v[i] = __shfl(a.v[i], src); return r; }
__device__ __noinline__ void tai_g2_group8(Fp2S& xo, Fp2S& yo, const u8* msg32, const u8* domain8, int g) {
    u32 w[16], dre[8], dim[8];
    for (int tag = 1; tag <= 2; tag++) {                                   // SHA-256 of the 41-byte string m || domain || tag
        for (int i = 0; i < 8; i++) w[i] = ((u32)msg32[4 * i] << 24) | ((u32)msg32[4 * i + 1] << 16) | ((u32)msg32[4 * i + 2] << 8) | msg32[4 * i + 3];
        for (int i = 0; i < 2; i++) w[8 + i] = ((u32)domain8[4 * i] << 24) | ((u32)domain8[4 * i + 1] << 16) | ((u32)domain8[4 * i + 2] << 8) | domain8[4 * i + 3];
        w[10] = ((u32)tag << 24) | 0x800000;
        w[11] = w[12] = w[13] = w[14] = 0;
        w[15] = 41 * 8;
        u32* d = tag == 1 ? dre : dim;
        sha256_init(d);
        sha256_block(d, w);
    }
    u32 wre[12], wim[12];
    for (int j = 0; j < 8; j++) { wre[j] = dre[7 - j]; wim[j] = dim[7 - j]; }
    for (int j = 8; j < 12; j++) { wre[j] = 0; wim[j] = 0; }
    Fp2S xc; xc.c0 = fp_from_words(wre); xc.c1 = fp_from_words(wim);
    Fp2S one = fp2_one(), step = fp2_zero();
    for (int k = 0; k < 8; k++) {                                          // x0 + g, and the stride 8 of a round
        if (k < g) xc = fp2_store(fp2_add(xc, one));
        step = fp2_store(fp2_add(step, one));
    }
    const int lane = (int)(threadIdx.x & 63), base = lane & ~7;
    Fp2S xsel = fp2_zero(), gsel = fp2_zero(); FpS ssel = fp_zero();
    i32 done = 0;
    while (true) {
        const Fp2S gx = fp2_store(fp2_add(fp2_mul(fp2_sqr(xc), xc), C_B2));
        bool ok; FpS nrm;
        const FpS s = fp2_norm_root(gx, nrm, ok);
        const unsigned long long pass = __ballot(ok ? 1 : 0);
        const u32 mine = (u32)(pass >> base) & 0xffu;                      // the eight verdicts of this message
        const int win = base + (mine ? __builtin_ctz(mine) : 0);           // smallest passing counter
        const i32 take = (mine != 0 && !done) ? -1 : 0;
        Fp2S xw, gw; xw.c0 = fp_from_lane(xc.c0, win); xw.c1 = fp_from_lane(xc.c1, win);
        gw.c0 = fp_from_lane(gx.c0, win); gw.c1 = fp_from_lane(gx.c1, win);
        const FpS sw = fp_from_lane(s, win);
        xsel = fp2_select(take, xw, xsel); gsel = fp2_select(take, gw, gsel); ssel = fp_select(take, sw, ssel);
        done |= take;
        if (__all(done != 0)) break;
        xc = fp2_store(fp2_add(xc, step));
    }
    Fp2S y = fp2_sqrt_from_norm_root(gsel, ssel);                          // either root: the choice follows
    const i32 y_gt = fp2_sign_is_neg(y);                                   // favour y with Parity() (g2.go:1074-1077)
    y = fp2_select(y_gt, y, fp2_store(fp2_neg(y)));
    xo = xsel; yo = y;
}
// The same search for the smallest calls: one workgroup of EIGHT WAVES per message, wave g tests candidate x0 + 8 r + g with every
// lane computing the same values and the norm's exponentiation running one limb per lane (fp_row.cuh); the verdicts and the winner's
// (x, g(x), norm root) meet in LDS, wave 0 finishes the square root (a second wave-wide exponentiation) and stores.
// lds: 8 verdict words + 75 words of the winner.
__device__ __noinline__ void tai_g2_waves8(Fp2S& xo, Fp2S& yo, const u8* msg32, const u8* domain8, i32* lds) {
    u32 w[16], dre[8], dim[8];
    for (int tag = 1; tag <= 2; tag++) {                                   // SHA-256 of the 41-byte string m || domain || tag
        for (int i = 0; i < 8; i++) w[i] = ((u32)msg32[4 * i] << 24) | ((u32)msg32[4 * i + 1] << 16) | ((u32)msg32[4 * i + 2] << 8) | msg32[4 * i + 3];
        for (int i = 0; i < 2; i++) w[8 + i] = ((u32)domain8[4 * i] << 24) | ((u32)domain8[4 * i + 1] << 16) | ((u32)domain8[4 * i + 2] << 8) | domain8[4 * i + 3];
        w[10] = ((u32)tag << 24) | 0x800000;
        w[11] = w[12] = w[13] = w[14] = 0;
        w[15] = 41 * 8;
        u32* d = tag == 1 ? dre : dim;
        sha256_init(d);
        sha256_block(d, w);
    }
    u32 wre[12], wim[12];
    for (int j = 0; j < 8; j++) { wre[j] = dre[7 - j]; wim[j] = dim[7 - j]; }
    for (int j = 8; j < 12; j++) { wre[j] = 0; wim[j] = 0; }
    Fp2S xc; xc.c0 = fp_from_words(wre); xc.c1 = fp_from_words(wim);
    const int g = (int)(threadIdx.x >> 6);                                 // this wave's candidate within a round
    Fp2S one = fp2_one(), step = fp2_zero();
    for (int k = 0; k < 8; k++) {
        if (k < g) xc = fp2_store(fp2_add(xc, one));
        step = fp2_store(fp2_add(step, one));
    }
    while (true) {
        const Fp2S gx = fp2_store(fp2_add(fp2_mul(fp2_sqr(xc), xc), C_B2));
        bool ok; FpS nrm;
        const FpS s = fp2_norm_root<true>(gx, nrm, ok);
        __syncthreads();                                                   // the previous round's verdicts have been read
        if ((threadIdx.x & 63) == 0) lds[g] = ok ? 1 : 0;
        __syncthreads();
        int win = -1;
        for (int k = 7; k >= 0; k--) if (lds[k]) win = k;                  // smallest passing counter of the round
        if (win == g && (threadIdx.x & 63) == 0) {
            for (int i = 0; i < NL; i++) { lds[8 + i] = xc.c0.v[i]; lds[8 + NL + i] = xc.c1.v[i]; lds[8 + 2 * NL + i] = gx.c0.v[i]; lds[8 + 3 * NL + i] = gx.c1.v[i]; lds[8 + 4 * NL + i] = s.v[i]; }
        }
        if (win >= 0) break;                                               // uniform over the workgroup
        xc = fp2_store(fp2_add(xc, step));
    }
    __syncthreads();
    Fp2S xsel, gsel; FpS ssel;
    for (int i = 0; i < NL; i++) { xsel.c0.v[i] = lds[8 + i]; xsel.c1.v[i] = lds[8 + NL + i]; gsel.c0.v[i] = lds[8 + 2 * NL + i]; gsel.c1.v[i] = lds[8 + 3 * NL + i]; ssel.v[i] = lds[8 + 4 * NL + i]; }
    Fp2S y = fp2_sqrt_from_norm_root<true>(gsel, ssel);                    // every wave the same (only wave 0's result is stored)
    const i32 y_gt = fp2_sign_is_neg(y);                                   // favour y with Parity() (g2.go:1074-1077)
    y = fp2_select(y_gt, y, fp2_store(fp2_neg(y)));
    xo = xsel; yo = y;
}
}  // namespace blsmi
