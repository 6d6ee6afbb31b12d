// k_msm_pair.hip -- the G2 passes of the bucket-method MSM (msm.inc) in the LANE-PAIR layout: bucket accumulation, the running-sum
// pass and its folds, a lane pair per bucket / chunk, two waves per SIMD.  A translation unit of its own because these kernels are
// compiled with the lane-pair multiply cores as assembly blobs (BLSMI_ASM_CORES, core_asm.inc, as k_fe_pair.hip):
// k_g2_msm_bucket_raw_pair 12.2 -> 11.7 ms at 2^20 points, the 2^20-point G2 MSM 16.4 -> 15.6 ms; the scalar-multiplication kernels of
// k_curve.hip keep the function cores (99.5 -> 100.8 ms with blobs).
#ifndef BLSMI_LIMBS28                // (with the 14-limb blobs clang 22's machine scheduler segfaults on jac_add: function cores in that build)
#define BLSMI_ASM_CORES
#endif
#define BLSMI_SOA_NATIVE                // the MSM's SoA buffers are shared with k_curve.hip only (same limbs there)
#include "tower.cuh"
#include "device_io.cuh"
#include "glv.cuh"
#include "msm_raw.cuh"
#include "pair_field.cuh"
namespace P2 = blsmi::pairl;
#include "pair_point_io.inc"

// G2 bucket accumulation of the MSM (msm.inc step 2) with a lane PAIR per bucket: half the point state per lane, two waves per
// SIMD instead of one.  The record a pair leaves is the one-lane kernel's (jac_soa_store): coordinate c_par by lane par.
__global__ void __launch_bounds__(WG, 2) k_g2_msm_bucket_pair(const u8* pts, const u32* idx, const u32* offs, const u32* hist, const u32* perm, i32* buckets, size_t n, int c, size_t nb) {
    const int par = threadIdx.x & 1;
    const size_t j = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t jj = j < nb ? j : nb - 1;
    const size_t t = perm[jj];
    const u32 cnt = j < nb ? hist[t] : 0;
    const u32* slice = idx + (t >> c) * n + offs[t];
    P2::G2JacP acc = jac_zero<P2::Fp2S>();
    for (u32 k = 0; k < cnt; k++) acc = jac_add_affine(acc, pair_load_g2(pts + (size_t)192 * slice[k], par));
    if (j < nb) {
        soa_store(buckets, nb, t, 0 + par, acc.x.c); soa_store(buckets, nb, t, 2 + par, acc.y.c); soa_store(buckets, nb, t, 4 + par, acc.z.c);
        if (!par) buckets[(size_t)6 * NL * nb + t] = acc.inf;
    }
}
// G2 bucket accumulation over the raw-limb items of the endomorphism MSM (msm.inc): a lane pair per bucket, lane `par` loads its
// own coefficient of the item's variant (P, -psi P, psi^2 P, -psi^3 P).
BLSMI_DEV P2::G2AffP raw_item_g2_pair(const i32* raw, u32 item, int par) {
    const i32* o = raw + (size_t)RAW2_WORDS * (item & 0x3fffffffu) + (size_t)(item >> 30) * 4 * NL + par * NL;
    P2::G2AffP a;
    a.x = P2::wrap(raw_load(o)); a.y = P2::wrap(raw_load(o + 2 * NL));
    a.inf = raw[(size_t)RAW2_WORDS * (item & 0x3fffffffu) + 240];
    return a;
}
__global__ void __launch_bounds__(WG, 2) k_g2_msm_bucket_raw_pair(const i32* raw, const u32* idx, const u32* offs, const u32* hist, const u32* perm, i32* buckets, size_t per_win, size_t nb) {
    const int par = threadIdx.x & 1;
    const size_t j = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t jj = j < nb ? j : nb - 1;
    const size_t t = perm[jj];
    const u32 cnt = j < nb ? hist[t] : 0;
    const u32* slice = idx + (t >> 16) * per_win + offs[t];
    Xyzz<P2::Fp2S> xacc = xyzz_zero<P2::Fp2S>();
    // (no software prefetch here: with the gathers served from cache the kernel is 5 % faster at most -- issue-bound -- and the 28 registers of
    // a pending point bring 27 spills back)
    for (u32 k = 0; k < cnt; k++) xyzz_acc_affine(xacc, raw_item_g2_pair(raw, slice[k], par));        // XYZZ accumulator (curve.cuh), inlined, special cases branched out
    const P2::G2JacP acc = xyzz_to_jac(xacc);
    if (j < nb) {
        soa_store(buckets, nb, t, 0 + par, acc.x.c); soa_store(buckets, nb, t, 2 + par, acc.y.c); soa_store(buckets, nb, t, 4 + par, acc.z.c);
        if (!par) buckets[(size_t)6 * NL * nb + t] = acc.inf;
    }
}
// The running-sum pass and the fold of the MSM for G2 in the lane-pair layout (the one-lane k_g2_msm_chunk runs one wave per SIMD with
// 494 spilled registers, one VALU instruction per 13.7 cycles: profiles/r03a): a lane pair per chunk / per pair of chunk sums.
__global__ void __launch_bounds__(WG, 2) k_g2_msm_chunk_pair(const i32* buckets, i32* chunks, int c, int K, size_t nb, size_t nchunks_total) {
    const int par = threadIdx.x & 1;
    const size_t t0 = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t t = t0 < nchunks_total ? t0 : nchunks_total - 1;         // both lanes of a pair stay active
    const size_t per_win = ((size_t)1 << c) / K;
    const size_t w = t / per_win, j = t % per_win;
    const size_t lo = j * K;
    P2::G2JacP running = jac_zero<P2::Fp2S>(), local = jac_zero<P2::Fp2S>();
#pragma unroll 1
    for (int k = K - 1; k >= 0; k--) {
        running = jac_add_i(running, pair_soa_load(buckets, nb, (w << c) + lo + k, par));
        local = jac_add_i(local, running);
    }
    if (j == 0) local = jac_add(local, jac_neg(running));                  // lo - 1 = -1
    else local = jac_add(local, jac_mul_u64_public(running, (u64)(lo - 1)));
    if (t0 < nchunks_total) pair_soa_store(chunks, nchunks_total, t, par, local);
}
__global__ void __launch_bounds__(WG, 2) k_g2_msm_fold_pair(const i32* src, i32* dst, size_t seg, size_t half, int nwin) {
    const int par = threadIdx.x & 1;
    const size_t t0 = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t total = half * nwin;
    const size_t t = t0 < total ? t0 : total - 1;
    const size_t w = t / half, j = t % half;
    P2::G2JacP r = pair_soa_load(src, seg * nwin, w * seg + j, par);
    if (j + half < seg) r = jac_add(r, pair_soa_load(src, seg * nwin, w * seg + j + half, par));
    if (t0 < total) pair_soa_store(dst, half * nwin, t, par, r);
}
// lane-pair forms of the multiplication-free running-sum pass and of its fold (msm.inc)
#ifndef BLSMI_CHUNK_WAVES
#define BLSMI_CHUNK_WAVES 1         // 2^15 chunk pairs = 1 024 waves: one per SIMD anyway, so the whole register file (msm.inc: k_g1_msm_chunk2)
#endif
__global__ void __launch_bounds__(WG, BLSMI_CHUNK_WAVES) k_g2_msm_chunk2_pair(const i32* buckets, i32* out, int c, int K, size_t nb, size_t nct) {
    const int par = threadIdx.x & 1;
    const size_t t0 = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t t = t0 < nct ? t0 : nct - 1;
    const size_t per_win = ((size_t)1 << c) / K;
    const size_t w = t / per_win, j = t % per_win;
    const size_t lo = j * K;
    P2::G2JacP running = jac_zero<P2::Fp2S>(), local = jac_zero<P2::Fp2S>();
#pragma unroll 1
    for (int k = K - 1; k >= 0; k--) {
        running = jac_add_i(running, pair_soa_load(buckets, nb, (w << c) + lo + k, par));
        local = jac_add_i(local, running);
    }
    if (t0 < nct) { pair_soa_store(out, 2 * nct, t, par, running); pair_soa_store(out, 2 * nct, nct + t, par, local); }
}
__global__ void __launch_bounds__(WG, 2) k_g2_msm_fold2_pair(const i32* src, i32* dst, int narr, int nwin, size_t len, int io) {
    const int par = threadIdx.x & 1;
    const size_t half = len / 2, per_arr = (size_t)nwin * half, total = (size_t)(narr + 1) * per_arr;
    const size_t t0 = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t t = t0 < total ? t0 : total - 1;
    const size_t a = t / per_arr, r = t % per_arr, w = r / half, j = r % half;
    const size_t nsrc = (size_t)narr * nwin * len;
    // (uniform per pair; a wave mixes both kinds only at an array boundary)
    P2::G2JacP v;
    if (a < (size_t)narr) {
        const size_t base = (a * nwin + w) * len + 2 * j;
        v = jac_add(pair_soa_load(src, nsrc, base, par), pair_soa_load(src, nsrc, base + 1, par));
    } else {
        v = pair_soa_load(src, nsrc, (size_t)w * len + 2 * j + 1, par);
    }
    if (t0 < total) {
        if (io) {                                                          // the last level: the tail program of k_lat.hip reads the inter-kernel form
            soa_store_io(dst, total, t, 0 + par, v.x.c); soa_store_io(dst, total, t, 2 + par, v.y.c); soa_store_io(dst, total, t, 4 + par, v.z.c);
            if (!par) dst[(size_t)6 * NL_IO * total + t] = v.inf;
        } else pair_soa_store(dst, total, t, par, v);
    }
}
