// k_prepared_pair.hip -- public keys kept PREPARED in HBM (the reference's G2Prepared, g2.go:639-801, as a resident table).
//
// bls.MillerLoop takes its G2 arguments prepared (pairing.go:4-14: MillerLoopItem{P *G1Affine, Q *G2Prepared}): the 68 line
// coefficient triples of Q depend on Q alone, and g2pubs.Verify recomputes them for the same public key on every call
// (CompareTwoPairings -> G2AffineToPrepared, pairing.go:140-147).  A verifier that sees the same keys again and again --
// a validator set -- can pay for them once: blsmi_g2_prepare_batch_dev writes one 24 KB table per key (288 GB of HBM3E
// hold ten million), and the Miller loops below READ a key's lines instead of running its doubling / addition steps:
// the point arithmetic of the variable pair (7.7 k of the 34.6 k instructions of a loop iteration) disappears.
//
// One table (PREP_WORDS i32):
//   [0, 6120)      68 lines x 3 coefficients x (c0 | c1) x 15 limbs -- the layout of the start-up table of the G2 generator
//                  ([line][coefficient][c0|c1][limb], fp2_table_load), filled by the REFERENCE's steps (doubling_step /
//                  addition_step): the table IS the reference's G2Prepared.coeffs, in the device's limb form
//                  (k_prepared_export converts it back for the parity tests);
//   [6120, 6168)   the key's 192-byte affine record as it came in (small batches take the latency programs, which want the point);
//   [6168]         1 if the record was all zero (the point at infinity: MillerLoop panics upstream, the tuple's verdict is 0 here).
#include "pairing.cuh"
#include "device_io.cuh"
#include "prepared.h"

#define KERNEL_PAIR __global__ void __launch_bounds__(WG, 2)
namespace P2 = blsmi::pairl;
namespace {
struct PairG1 { FpS x, y; };
struct PairG2 { P2::Fp2S x, y; };
constexpr int PT = WG / 2;
constexpr int PREP_KEY_AT = blsmi_prep::KEY_AT, PREP_FLAG_AT = blsmi_prep::FLAG_AT, PREP_WORDS = blsmi_prep::WORDS;
static_assert(blsmi_prep::LINE_WORDS == 68 * 3 * 2 * NL_IO, "table layout: NL_IO = 15 words per coefficient, of which this build's limbs fill the first NL");

BLSMI_DEV void pair_store12(i32* buf, size_t n, size_t t, int par, const P2::Fp12S& f) {
    const FpS* c = reinterpret_cast<const FpS*>(&f);
#pragma unroll
    for (int j = 0; j < 6; j++) soa_store(buf, n, t, 2 * j + par, c[j]);
}
BLSMI_DEV const i32* table_of(const i32* tables, const u32* key_idx, size_t t) {
    return tables + (size_t)(key_idx ? key_idx[t] : t) * PREP_WORDS;
}
}  // namespace

// G2AffineToPrepared (g2.go:650-801) of n keys, one lane pair per key
KERNEL_PAIR k_g2_prepare_pair(const u8* g2, i32* tables, size_t n) {
    const int par = threadIdx.x & 1;
    const size_t t = (size_t)blockIdx.x * PT + (threadIdx.x >> 1);
    const size_t tt = t < n ? t : n - 1;                                   // idle lane pairs redo the last key (identical stores)
    const u8* rec = g2 + 192 * tt;
    i32* table = tables + tt * PREP_WORDS;
    P2::prepare_lines(P2::wrap(load_be48(rec + 48 * par)), P2::wrap(load_be48(rec + 96 + 48 * par)), table);
    const u32* w = reinterpret_cast<const u32*>(rec);
    u32 any = 0;
    for (int i = 0; i < 24; i++) { const u32 v = w[24 * par + i]; table[PREP_KEY_AT + 24 * par + i] = (i32)v; any |= v; }
    any |= __shfl_xor((int)any, 1);
    if (par == 0) table[PREP_FLAG_AT] = any ? 0 : 1;
}
// the reference's G2Prepared.coeffs of keys [0, n): 68 x 3 Fq2, each c0 | c1 as 6 x u64 Montgomery limbs (R = 2^384)
__global__ void __launch_bounds__(WG) k_prepared_export(const i32* tables, u64* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * WG + threadIdx.x;               // one Fq per lane
    if (i >= n * 408) return;
    const size_t key = i / 408; const int e = (int)(i % 408);
    FpS x;
    const i32* src = tables + key * PREP_WORDS + (size_t)e * NL_IO;
#pragma unroll
    for (int k = 0; k < NL; k++) x.v[k] = src[k];
    store_m384(out + 6 * i, x);
}
// flags[t] = in_flags[t] | (key of tuple t is the zero record) | (signature record t is all zero)
// *any_flag (may be null) is raised when some tuple is flagged
__global__ void __launch_bounds__(WG) k_flag_prepared(const i32* tables, const u32* key_idx, const u8* sigs, int sig_words, const u8* in_flags, u8* flags, int* any_flag, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    u32 any = 0;
    if (sigs) { const u32* s = reinterpret_cast<const u32*>(sigs) + t * sig_words; for (int i = 0; i < sig_words; i++) any |= s[i]; } else any = 1;
    const u8 fl = (u8)((in_flags && in_flags[t]) || !any || table_of(tables, key_idx, t)[PREP_FLAG_AT]);
    flags[t] = fl;
    if (fl && any_flag) atomicOr(any_flag, 1);
}
// pks[t] = the affine record inside tuple t's table (the latency programs take points)
__global__ void __launch_bounds__(WG) k_prepared_gather_keys(const i32* tables, const u32* key_idx, u32* pks, size_t n) {
    const size_t i = (size_t)blockIdx.x * WG + threadIdx.x;
    if (i >= n * 48) return;
    pks[i] = (u32)table_of(tables, key_idx, i / 48)[PREP_KEY_AT + (int)(i % 48)];
}

// MillerLoop(P_t, prepared Q_t), one pairing per lane pair (Pairing with a prepared second argument)
KERNEL_PAIR k_miller1_prep_pair(const u8* g1, const i32* tables, const u32* key_idx, i32* fbuf, size_t n) {
    const int par = threadIdx.x & 1;
    const size_t t = (size_t)blockIdx.x * PT + (threadIdx.x >> 1);
    const size_t tt = t < n ? t : n - 1;
    PairG1 p[1]; PairG2 q[1];
    p[0].x = load_be48(g1 + 96 * tt); p[0].y = load_be48(g1 + 96 * tt + 48);
    q[0].x = P2::fp2_one(); q[0].y = P2::fp2_one();                       // not read
    P2::Fp12S f;
    P2::miller_loop<1, true, false>(f, p, q, table_of(tables, key_idx, tt));
    if (t < n) pair_store12(fbuf, n, t, par, f);
}
// g2pubs.Verify with a prepared key: MillerLoop((sig_t, G2 generator), (-H(m_t), pk_t)) -- BOTH pairs read their lines
// (the generator's from the start-up table, the key's from its own), so the loop is 63 x (two lines multiplied together, times f, squared)
KERNEL_PAIR k_miller2_prep_pair(const u8* sigs, const u8* h, const i32* tables, const u32* key_idx, i32* fbuf, size_t n, const i32* pre_gen) {
    const int par = threadIdx.x & 1;
    const size_t t = (size_t)blockIdx.x * PT + (threadIdx.x >> 1);
    const size_t tt = t < n ? t : n - 1;
    PairG1 p[2]; PairG2 q[2];
    p[0].x = load_be48(sigs + 96 * tt); p[0].y = load_be48(sigs + 96 * tt + 48);
    p[1].x = load_be48(h + 96 * tt); p[1].y = fp_store(fp_neg(load_be48(h + 96 * tt + 48)));
    q[0].x = P2::fp2_one(); q[0].y = P2::fp2_one(); q[1] = q[0];
    P2::Fp12S f;
    P2::miller_loop<2, true, false, true>(f, p, q, pre_gen, table_of(tables, key_idx, tt));
    if (t < n) pair_store12(fbuf, n, t, par, f);
}
// The Miller loops of a VerifyAggregate over prepared keys: two consecutive tuples per lane pair as in k_miller1x2_pair
KERNEL_PAIR k_miller1x2_prep_pair(const u8* g1, const i32* tables, const u32* key_idx, i32* fbuf, size_t n, size_t m) {
    const int par = threadIdx.x & 1;
    const size_t t = (size_t)blockIdx.x * PT + (threadIdx.x >> 1);
    const size_t tt = t < m ? t : m - 1;
    const size_t i0 = 2 * tt, i1 = 2 * tt + 1 < n ? 2 * tt + 1 : i0;
    PairG1 p[2]; PairG2 q[2];
    p[0].x = load_be48(g1 + 96 * i0); p[0].y = load_be48(g1 + 96 * i0 + 48);
    p[1].x = load_be48(g1 + 96 * i1); p[1].y = load_be48(g1 + 96 * i1 + 48);
    q[0].x = P2::fp2_one(); q[0].y = P2::fp2_one(); q[1] = q[0];
    const i32* ta = table_of(tables, key_idx, i0);
    const i32* tb = table_of(tables, key_idx, i1);
    P2::Fp12S f;
    if (__all(2 * tt + 1 < n)) P2::miller_loop<2, true, false, true>(f, p, q, ta, tb);
    else {                                                                 // the wave that holds the odd tuple out
        P2::Fp12S f2, f1;
        P2::miller_loop<2, true, false, true>(f2, p, q, ta, tb);
        PairG1 p1[1] = {p[0]}; PairG2 q1[1] = {q[0]};
        P2::miller_loop<1, true, false>(f1, p1, q1, ta);
        const i32 both = 2 * tt + 1 < n ? -1 : 0;
        FpS* a = reinterpret_cast<FpS*>(&f2); const FpS* b = reinterpret_cast<const FpS*>(&f1);
        for (int j = 0; j < 6; j++) a[j] = fp_select(both, a[j], b[j]);
        f = f2;
    }
    if (t < m) pair_store12(fbuf, m, t, par, f);
}
