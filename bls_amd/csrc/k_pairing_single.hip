// k_pairing_single.hip -- Miller-loop kernels in the one-tuple-per-lane layout (BLSMI_LAYOUT=single), the 2-pair Miller loop of
// the verify path in that layout, and the start-up preparation of the generator's line table.  The final-exponentiation side of
// the same layout lives in k_fe_single.hip (two translation units compile side by side: this one alone took 77 s).
#include "pairing.cuh"
#include "device_io.cuh"

// ------------------------------------------------------------------------------------------------
// kernels: pairing
// ------------------------------------------------------------------------------------------------
// Miller loop for one pair per tuple; f goes to the internal SoA buffer (or nowhere else).
template <bool EXACT> BLSMI_DEV void miller1_body(const u8* g1, const u8* g2, i32* fbuf, size_t n) {
    __shared__ u32 lds[WG * 49];
    const size_t first = (size_t)blockIdx.x * WG;
    const size_t t = first + threadIdx.x;
    const int rec = (t < n) ? (int)threadIdx.x : (int)(n - 1 - first);   // tail lanes redo the last tuple
    G1Aff p[1]; G2Aff q[1];
    tile_load<24>(lds, g1, first, n);
    p[0] = lds_g1(lds + rec * 25);
    __syncthreads();
    tile_load<48>(lds, g2, first, n);
    q[0] = lds_g2(lds + rec * 49);
    Fp12S f;
    miller_loop<1, false, EXACT>(f, p, q);
    if (t < n) soa_store12(fbuf, n, t, f);
}
// k_miller1: the reference's Miller value; k_miller1h: the homogeneous steps (see pairing_body.inc)
KERNEL k_miller1(const u8* g1, const u8* g2, i32* fbuf, size_t n) { miller1_body<true>(g1, g2, fbuf, n); }
KERNEL k_miller1h(const u8* g1, const u8* g2, i32* fbuf, size_t n) { miller1_body<false>(g1, g2, fbuf, n); }
// G2Prepared of the generator (g2.go:639-801): 68 line-coefficient triples for the fixed-Q Miller loop of g2pubs.Verify
KERNEL k_prepare_generator_lines(const u8* g2, i32* table) {
    if (threadIdx.x != 0) return;
    const G2Aff q = load_g2(g2);
    prepare_lines(q.x, q.y, table);
}
// ---- CompareTwoPairings (pairing.go:140-147): f = ML((P0,Q0), (-P1,Q1)); strides in bytes, 0 = broadcast
// `pre` != nullptr: Q0 is the G2 generator and its line coefficients come from the start-up table
KERNEL k_miller2(const u8* p0, size_t sp0, const u8* q0, size_t sq0, const u8* p1, size_t sp1, const u8* q1, size_t sq1, i32* fbuf, size_t n, const i32* pre) {
    __shared__ u32 lds[WG * 49];
    const size_t first = (size_t)blockIdx.x * WG;
    const size_t t = first + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    const int rec = (t < n) ? (int)threadIdx.x : (int)(n - 1 - first);
    G1Aff p[2]; G2Aff q[2];
    // stride 0 = one broadcast record (a generator); otherwise stage the wave's 64 records through LDS
    if (sp0) { tile_load<24>(lds, p0, first, n); p[0] = lds_g1(lds + rec * 25); __syncthreads(); } else p[0] = load_g1(p0);
    if (sq0) { tile_load<48>(lds, q0, first, n); q[0] = lds_g2(lds + rec * 49); __syncthreads(); } else q[0] = load_g2(q0);
    tile_load<24>(lds, p1, first, n); p[1] = aff_neg(lds_g1(lds + rec * 25)); __syncthreads();
    tile_load<48>(lds, q1, first, n); q[1] = lds_g2(lds + rec * 49);
    (void)tt; (void)sp1; (void)sq1;
    Fp12S f;
    if (pre) miller_loop<2, true, false>(f, p, q, pre);
    else miller_loop<2, false, false>(f, p, q);
    if (t < n) soa_store12(fbuf, n, t, f);
}
// G2AffineToPrepared (g2.go:650-801) of one point, by the doubling / addition steps of either lane layout, into the
// [line][coefficient][c0|c1][limb] table format of the fixed-generator Miller loop; and that table -> 6 x u64 Montgomery-384
KERNEL k_debug_prepare_single(const u8* g2, i32* table) { if (threadIdx.x == 0) { const G2Aff q = load_g2(g2); prepare_lines(q.x, q.y, table); } }
KERNEL k_debug_lines_to_m384(const i32* table, u64* out) {
    const int e = blockIdx.x * WG + threadIdx.x;                          // one Fq per lane: 68 lines x 3 coefficients x 2
    if (e >= 68 * 3 * 2) return;
    FpS v;
    for (int i = 0; i < NL; i++) v.v[i] = table[e * NL + i];
    store_m384(out + 6 * e, v);
}
