// k_curve.hip -- scalar multiplication, point sums, the bucket-method MSM, and the Fq6 / curve unit ops of the parity tests.
#define BLSMI_SOA_NATIVE                // the Jacobian SoA buffers of the sums and of the MSM passes are read by this unit (and k_msm_pair.hip) only
#include "tower.cuh"
#include "device_io.cuh"
#include "glv.cuh"
#include "msm_raw.cuh"

KERNEL k_debug_fq6(int op, const u64* a, const u64* b, u64* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<6> ra = rec_load<6>(a, t), rb = ra, ro;
    if (op == BLSMI_OP_FQ6_MUL || op == BLSMI_OP_FQ6_MUL_BY_1 || op == BLSMI_OP_FQ6_MUL_BY_01) rb = rec_load<6>(b, t);
    const Fp6S x = as<Fp6S>(ra), y = as<Fp6S>(rb);
    Fp6S r = fp6_zero();
    switch (op) {
        case BLSMI_OP_FQ6_MUL_BY_1: r = fp6_store(fp6_mul_by_1(x, y.c0)); break;                    // fq6.go:40-57, c1 = first Fq2 of b
        case BLSMI_OP_FQ6_MUL_BY_01: r = fp6_store(fp6_mul_by_01(x, y.c0, y.c1)); break;            // fq6.go:60-90, (c0, c1) = first two Fq2 of b
        case BLSMI_OP_FQ6_MUL: r = fp6_store(fp6_mul(x, y)); break;
        case BLSMI_OP_FQ6_SQR: r = fp6_store(fp6_sqr(x)); break;
        case BLSMI_OP_FQ6_INV: r = fp6_store(fp6_inv(x)); break;
        case BLSMI_OP_FQ6_FROB1: r = fp6_store(fp6_frob<1>(x)); break;
    }
    as<Fp6S>(ro) = r;
    rec_store<6>(out, t, ro);
}
// Jacobian points as 3 (G1) / 6 (G2) Fq records x,y,z; infinity <=> z == 0 (g1.go:293)
template <class F, int W>
BLSMI_DEV void debug_curve(int dbl, const u64* a, const u64* b, u64* out, size_t t) {
    Rec<W> ra = rec_load<W>(a, t), rb = ra, ro;
    if (!dbl) rb = rec_load<W>(b, t);
    Jac<F> p, q, r;
    p.x = reinterpret_cast<F*>(&ra)[0]; p.y = reinterpret_cast<F*>(&ra)[1]; p.z = reinterpret_cast<F*>(&ra)[2]; p.inf = f_is_zero(p.z) ? -1 : 0;
    q.x = reinterpret_cast<F*>(&rb)[0]; q.y = reinterpret_cast<F*>(&rb)[1]; q.z = reinterpret_cast<F*>(&rb)[2]; q.inf = f_is_zero(q.z) ? -1 : 0;
    r = dbl ? jac_double(p) : jac_add(p, q);
    if (r.inf) r.z = field_consts<F>::zero();
    reinterpret_cast<F*>(&ro)[0] = r.x; reinterpret_cast<F*>(&ro)[1] = r.y; reinterpret_cast<F*>(&ro)[2] = r.z;
    rec_store<W>(out, t, ro);
}
KERNEL k_debug_curve(int op, const u64* a, const u64* b, u64* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    if (op == BLSMI_OP_G1_DOUBLE || op == BLSMI_OP_G1_ADD) debug_curve<FpS, 3>(op == BLSMI_OP_G1_DOUBLE, a, b, out, t);
    else debug_curve<Fp2S, 6>(op == BLSMI_OP_G2_DOUBLE, a, b, out, t);
}
// ------------------------------------------------------------------------------------------------
// kernels: scalar multiplication, sums
// ------------------------------------------------------------------------------------------------
template <class F> BLSMI_DEV Aff<F> load_aff(const u8* p);
template <> BLSMI_DEV G1Aff load_aff<FpS>(const u8* p) { return load_g1(p); }
template <> BLSMI_DEV G2Aff load_aff<Fp2S>(const u8* p) { return load_g2(p); }
BLSMI_DEV void store_aff(u8* p, const G1Aff& a) { store_g1(p, a); }
BLSMI_DEV void store_aff(u8* p, const G2Aff& a) { store_g2(p, a); }

// Fixed 4-bit-window scalar multiplication (BASELINE config 3).  The reference multiplies bit-serially
// (g1.go:80-90, g2.go:92-102: 255 doublings + one addition per set bit); here each lane builds the table
// {0, P, 2P, ..., 15P} in its scratch (per-lane indexed), then per nibble does four doublings and ONE addition:
// 252 doublings + 63 + 14 additions, uniform control flow for all 64 lanes.  Same group element, so the
// affine output is identical to the reference's.
template <class F, int PB>
__device__ void mul_batch_body(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    const Aff<F> p = load_aff<F>(pts + pt_stride * tt);                  // stride 0: one common base point (PrivToPub)
    const u32* s32 = reinterpret_cast<const u32*>(scalars + 32 * tt);
    Jac<F> tab[16];
    tab[0] = jac_zero<F>();
    tab[1] = to_jac(p);
    for (int j = 2; j < 16; j++) tab[j] = jac_add_affine(tab[j - 1], p);
    Jac<F> res = jac_zero<F>();
    for (int w = 0; w < 8; w++) {                                        // big-endian scalar: word 0 is most significant
        const u32 kw = __builtin_bswap32(s32[w]);
        for (int nib = 7; nib >= 0; nib--) {
            if (w | (7 - nib)) { res = jac_double(res); res = jac_double(res); res = jac_double(res); res = jac_double(res); }
            res = jac_add(res, tab[(kw >> (4 * nib)) & 15]);
        }
    }
    const Aff<F> a = jac_to_affine(res);
    if (t < n) { store_aff(out + (size_t)PB * t, a); out_inf[t] = a.inf ? 1 : 0; }
}
// the verdict byte of a level program (1 = finite result) as the int32 infinity flag of the sum / MSM entry points
KERNEL k_good_to_flag(const u8* good, i32* flag) { if (blockIdx.x == 0 && threadIdx.x == 0) *flag = good[0] ? 0 : 1; }
// Small batches run the multiplication as a level program of the latency path (k_lat.hip: mul1 / mul2); this kernel turns its
// verdict into the out_inf byte and applies the library's convention for a multiplicand given as the all-zero record.
KERNEL k_mul_finish(const u8* good, const u8* pts, size_t pt_stride, int rec_words, u8* out, u8* out_inf, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    const u32* in = reinterpret_cast<const u32*>(pts + pt_stride * t);
    u32 any = 0;
    for (int i = 0; i < rec_words; i++) any |= in[i];
    const bool inf = !good[t] || any == 0;
    out_inf[t] = inf ? 1 : 0;
    if (inf) { u32* w = reinterpret_cast<u32*>(out) + (size_t)rec_words * t; for (int i = 0; i < rec_words; i++) w[i] = 0; }
}
KERNEL2 k_g1_mul(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_batch_body<FpS, 96>(pts, pt_stride, scalars, out, out_inf, n); }
KERNEL k_g2_mul(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_batch_body<Fp2S, 192>(pts, pt_stride, scalars, out, out_inf, n); }
// The same multiplications through the curve endomorphisms (glv.cuh): half (G1) / a quarter (G2) of the doublings.  For points of
// the prime-order subgroup -- the default of the scalar-multiplication entry points, see blsmi_set_mul_assume_subgroup.
template <class F, int PB>
__device__ void mul_glv_body(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    const Aff<F> p = load_aff<F>(pts + pt_stride * tt);
    const Aff<F> a = jac_to_affine(glv_mul<F>(p, scalars + 32 * tt));
    if (t < n) { store_aff(out + (size_t)PB * t, a); out_inf[t] = a.inf ? 1 : 0; }
}
// The digit record the scalar-multiplication level programs read (gen_lat.py build_mul_program, glv_model.lat_record_g*): 64
// big-endian bytes per scalar -- group 1: k1 | k2 << 256 with k = k1 + k2 z^2; group 2: sum d_i << 128 i with k = sum d_i z^i.
KERNEL k_glv_recode(const u8* scalars, int group, u8* rec, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    u32 k[8];
    scalar_words(scalars + 32 * t, k);
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
    if (group == 1) {
        u32 k2[5], k1[5];
        udivmod_recip<8, 4, 7, 5>(k, C_GLV_Z2, C_GLV_MZ2, k2, k1);
#pragma unroll
        for (int i = 0; i < 5; i++) { w[i] = k1[i]; w[8 + i] = k2[i]; }
    } else {
        u32 q1[7], q2[5], q3[3], r[3];
        udivmod_recip<8, 2, 9, 7>(k, C_GLV_Z, C_GLV_MZ, q1, r);
#pragma unroll
        for (int i = 0; i < 3; i++) w[i] = r[i];
        udivmod_recip<7, 2, 9, 5>(q1, C_GLV_Z, C_GLV_MZ, q2, r);
#pragma unroll
        for (int i = 0; i < 3; i++) w[4 + i] = r[i];
        udivmod_recip<5, 2, 9, 3>(q2, C_GLV_Z, C_GLV_MZ, q3, r);
#pragma unroll
        for (int i = 0; i < 3; i++) { w[8 + i] = r[i]; w[12 + i] = q3[i]; }
    }
    u32* o = reinterpret_cast<u32*>(rec + 64 * t);
#pragma unroll
    for (int i = 0; i < 16; i++) o[i] = __builtin_bswap32(w[15 - i]);
}
#ifndef BLSMI_GLV_WAVES
#define BLSMI_GLV_WAVES 2
#endif
__global__ void __launch_bounds__(WG, BLSMI_GLV_WAVES) k_g1_mul_glv(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_glv_body<FpS, 96>(pts, pt_stride, scalars, out, out_inf, n); }
KERNEL k_g2_mul_glv(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_glv_body<Fp2S, 192>(pts, pt_stride, scalars, out, out_inf, n); }

// ---- fixed-base multiplication of the group generators (PrivToPub, g2pubs/bls.go:138-140, g1pubs/bls.go:144-146) ------------------
// [k] G = sum_w T[w][byte_w(k)] with T[w][d] = [d 256^w] G for d = 1 .. 255, w = 0 .. 31: thirty-two mixed additions and no
// doubling at all (the windowed / endomorphism ladders spend 252 / 125 doublings).  The table is built once per device at start-up
// by the scalar-multiplication kernel itself (blsmi.hip init_device) and kept as raw limbs: entry (w, d) at word ((w 255 + d - 1) E)
// with E = 2 (G1) / 4 (G2) field elements of 15 words.  980 KB for G1, 1.9 MB for G2: resident in L2.
constexpr int FIXED_WINDOWS = 32, FIXED_ENTRIES = 255;
template <class F> BLSMI_DEV Aff<F> fixed_entry(const i32* table, int w, u32 d);
template <> BLSMI_DEV G1Aff fixed_entry<FpS>(const i32* table, int w, u32 d) {
    const i32* e = table + ((size_t)w * FIXED_ENTRIES + (d ? d - 1 : 0)) * 2 * NL;
    G1Aff a; a.x = raw_load(e); a.y = raw_load(e + NL); a.inf = d ? 0 : -1; return a;
}
template <> BLSMI_DEV G2Aff fixed_entry<Fp2S>(const i32* table, int w, u32 d) {
    const i32* e = table + ((size_t)w * FIXED_ENTRIES + (d ? d - 1 : 0)) * 4 * NL;
    G2Aff a; a.x.c0 = raw_load(e); a.x.c1 = raw_load(e + NL); a.y.c0 = raw_load(e + 2 * NL); a.y.c1 = raw_load(e + 3 * NL); a.inf = d ? 0 : -1; return a;
}
// affine wire records (what the multiplication kernel wrote) -> the raw-limb table
KERNEL k_fixed_table_from_wire(const u8* wire, int elems, i32* table, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n * elems) return;
    const FpS x = load_be48(wire + 48 * t);
    for (int j = 0; j < NL; j++) table[t * NL + j] = x.v[j];
}
// throughput form: one scalar per lane
template <class F, int PB>
__device__ void mul_fixed_body(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    const u8* s = scalars + 32 * tt;
    Jac<F> res = jac_zero<F>();
#pragma unroll 1
    for (int w = 0; w < FIXED_WINDOWS; w++) res = jac_add_affine_i(res, fixed_entry<F>(table, w, s[31 - w]));
    const Aff<F> a = jac_to_affine(res);
    if (t < n) { store_aff(out + (size_t)PB * t, a); out_inf[t] = a.inf ? 1 : 0; }
}
KERNEL2 k_g1_mul_fixed(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_fixed_body<FpS, 96>(table, scalars, out, out_inf, n); }
KERNEL k_g2_mul_fixed(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_fixed_body<Fp2S, 192>(table, scalars, out, out_inf, n); }
// latency form: one scalar per WAVE -- lane w fetches the entry of window w, the 32 points meet in a five-level tree of
// additions across the lanes (a point travels by wave shuffles), lane 0 converts to affine: 5 dependent additions + one inversion
// instead of 32 + one.
template <class F> BLSMI_DEV Jac<F> jac_shfl_down(const Jac<F>& p, int off) {
    Jac<F> r;
    const i32* src = reinterpret_cast<const i32*>(&p);
    i32* dst = reinterpret_cast<i32*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(Jac<F>) / 4); i++) dst[i] = __shfl_down(src[i], off);
    return r;
}
template <class F, int PB>
__device__ void mul_fixed_wave_body(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n) {
    const size_t t = blockIdx.x;
    const int lane = threadIdx.x;
    const int w = lane & (FIXED_WINDOWS - 1);
    Aff<F> e = fixed_entry<F>(table, w, scalars[32 * t + 31 - w]);
    if (lane >= FIXED_WINDOWS) e.inf = -1;
    Jac<F> acc = to_jac(e);
#pragma unroll 1
    for (int off = FIXED_WINDOWS / 2; off >= 1; off >>= 1) acc = jac_add(acc, jac_shfl_down(acc, off));
    const Aff<F> a = jac_to_affine(acc);
    if (lane == 0) { store_aff(out + (size_t)PB * t, a); out_inf[t] = a.inf ? 1 : 0; }
}
KERNEL2 k_g1_mul_fixed_wave(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_fixed_wave_body<FpS, 96>(table, scalars, out, out_inf, n); }
KERNEL k_g2_mul_fixed_wave(const i32* table, const u8* scalars, u8* out, u8* out_inf, size_t n) { mul_fixed_wave_body<Fp2S, 192>(table, scalars, out, out_inf, n); }

// Point sums: level 0 reads affine bytes pairwise into Jacobian SoA; later levels halve the array.
template <class F> struct jac_words { static constexpr int value = sizeof(F) / sizeof(FpS) * 3; };
template <class F>
BLSMI_DEV void jac_soa_store(i32* buf, size_t n, size_t t, const Jac<F>& p) {
    constexpr int W = jac_words<F>::value;
    const FpS* c = reinterpret_cast<const FpS*>(&p);
#pragma unroll
    for (int e = 0; e < W; e++) soa_store(buf, n, t, e, c[e]);
    buf[(size_t)W * NL * n + t] = p.inf;
}
// the same record in the inter-kernel form (15 x 27-bit limbs): what the MSM's last fold level leaves for the tail program of k_lat.hip
template <class F>
BLSMI_DEV void jac_soa_store_io(i32* buf, size_t n, size_t t, const Jac<F>& p) {
    constexpr int W = jac_words<F>::value;
    const FpS* c = reinterpret_cast<const FpS*>(&p);
#pragma unroll
    for (int e = 0; e < W; e++) soa_store_io(buf, n, t, e, c[e]);
    buf[(size_t)W * NL_IO * n + t] = p.inf;
}
template <class F>
BLSMI_DEV Jac<F> jac_soa_load(const i32* buf, size_t n, size_t t) {
    constexpr int W = jac_words<F>::value;
    Jac<F> p;
    FpS* c = reinterpret_cast<FpS*>(&p);
#pragma unroll
    for (int e = 0; e < W; e++) c[e] = soa_load(buf, n, t, e);
    p.inf = buf[(size_t)W * NL * n + t];
    return p;
}
template <class F, int PB>
__device__ void sum_level0_body(const u8* pts, const u8* in_inf, i32* buf, size_t n, size_t half) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= half) return;
    Aff<F> a = load_aff<F>(pts + (size_t)PB * t);
    if (in_inf && in_inf[t]) a.inf = -1;
    Jac<F> r = to_jac(a);
    if (t + half < n) {
        Aff<F> b = load_aff<F>(pts + (size_t)PB * (t + half));
        if (in_inf && in_inf[t + half]) b.inf = -1;
        r = jac_add_affine(r, b);
    }
    jac_soa_store(buf, half, t, r);
}
KERNEL2 k_g1_sum0(const u8* pts, const u8* in_inf, i32* buf, size_t n, size_t half) { sum_level0_body<FpS, 96>(pts, in_inf, buf, n, half); }
KERNEL k_g2_sum0(const u8* pts, const u8* in_inf, i32* buf, size_t n, size_t half) { sum_level0_body<Fp2S, 192>(pts, in_inf, buf, n, half); }
// level 0 over the reference's in-memory points (bls.G?Projective records, device_io.cuh: load_jac_m384; the *_jac entry points): the sum of
// Jacobian points needs no ToAffine at all -- g1.go:400-470 / g2.go:446-516 (AddAssign of two projective points), as AggregatePublicKeys runs it
template <class F>
__device__ void sum_level0_jac_body(const u64* pts, i32* buf, size_t n, size_t half) {
    constexpr int RW = 18 * (sizeof(F) / sizeof(FpS));
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= half) return;
    Jac<F> r = load_jac_m384<F>(pts + (size_t)RW * t);
    if (t + half < n) r = jac_add(r, load_jac_m384<F>(pts + (size_t)RW * (t + half)));
    jac_soa_store(buf, half, t, r);
}
KERNEL2 k_g1_sum0_jac(const u64* pts, i32* buf, size_t n, size_t half) { sum_level0_jac_body<FpS>(pts, buf, n, half); }
KERNEL k_g2_sum0_jac(const u64* pts, i32* buf, size_t n, size_t half) { sum_level0_jac_body<Fp2S>(pts, buf, n, half); }
template <class F>
__device__ void sum_level_body(const i32* src, i32* dst, size_t n, size_t half) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= half) return;
    Jac<F> r = jac_soa_load<F>(src, n, t);
    if (t + half < n) r = jac_add(r, jac_soa_load<F>(src, n, t + half));
    jac_soa_store(dst, half, t, r);
}
KERNEL2 k_g1_sum(const i32* src, i32* dst, size_t n, size_t half) { sum_level_body<FpS>(src, dst, n, half); }
KERNEL k_g2_sum(const i32* src, i32* dst, size_t n, size_t half) { sum_level_body<Fp2S>(src, dst, n, half); }
template <class F, int PB>
__device__ void sum_final_body(const i32* src, u8* out, i32* out_inf) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const Aff<F> a = jac_to_affine(jac_soa_load<F>(src, 1, 0));
    store_aff(out, a);
    *out_inf = a.inf ? 1 : 0;
}
KERNEL2 k_g1_sum_final(const i32* src, u8* out, i32* out_inf) { sum_final_body<FpS, 96>(src, out, out_inf); }
KERNEL k_g2_sum_final(const i32* src, u8* out, i32* out_inf) { sum_final_body<Fp2S, 192>(src, out, out_inf); }

#include "msm.inc"

// ------------------------------------------------------------------------------------------------------------------
// G2 scalar multiplication in the LANE-PAIR layout (pair_field.cuh): lanes 2k, 2k+1 share point k, one Fq2 coefficient
// per lane -- 45 registers of point state per lane instead of 90, the per-lane window table halves, and the kernel runs
// two waves per SIMD (the one-point-per-lane k_g2_mul needs the whole register file: one wave).  Same window method,
// same group element, same affine bytes.
// ------------------------------------------------------------------------------------------------------------------
#include "pair_field.cuh"
namespace P2 = blsmi::pairl;
#include "pair_point_io.inc"
__global__ void __launch_bounds__(WG, 2) k_g2_mul_pair(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) {
    const int par = threadIdx.x & 1;
    const size_t t = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t tt = t < n ? t : n - 1;
    const P2::G2AffP p = pair_load_g2(pts + pt_stride * tt, par);
    const u32* s32 = reinterpret_cast<const u32*>(scalars + 32 * tt);
    P2::G2JacP tab[16];
    tab[0] = jac_zero<P2::Fp2S>();
    tab[1] = to_jac(p);
    for (int j = 2; j < 16; j++) tab[j] = jac_add_affine(tab[j - 1], p);
    P2::G2JacP res = jac_zero<P2::Fp2S>();
    for (int w = 0; w < 8; w++) {
        const u32 kw = __builtin_bswap32(s32[w]);
        for (int nib = 7; nib >= 0; nib--) {
            if (w | (7 - nib)) { res = jac_double(res); res = jac_double(res); res = jac_double(res); res = jac_double(res); }
            res = jac_add(res, tab[(kw >> (4 * nib)) & 15]);
        }
    }
    const P2::G2AffP a = jac_to_affine(res);
    if (t < n) { pair_store_g2(out + (size_t)192 * t, par, a); if (!par) out_inf[t] = a.inf ? 1 : 0; }
}
__global__ void __launch_bounds__(WG, BLSMI_GLV_WAVES) k_g2_mul_glv_pair(const u8* pts, size_t pt_stride, const u8* scalars, u8* out, u8* out_inf, size_t n) {
    const int par = threadIdx.x & 1;
    const size_t t = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t tt = t < n ? t : n - 1;
    const P2::G2AffP p = pair_load_g2(pts + pt_stride * tt, par);
    const P2::G2AffP a = jac_to_affine(glv_mul<P2::Fp2S>(p, scalars + 32 * tt));
    if (t < n) { pair_store_g2(out + (size_t)192 * t, par, a); if (!par) out_inf[t] = a.inf ? 1 : 0; }
}
