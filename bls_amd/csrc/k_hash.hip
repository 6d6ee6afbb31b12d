// k_hash.hip -- hash-to-curve kernels, the compressed wire format, and the Fq / Fq2-level unit ops of the parity tests.
#include "hash.cuh"
#include "device_io.cuh"

// verify_kernels.inc -- kernels behind the g1pubs / g2pubs verify surface, the hash-to-curve batch
// entry points and the compressed wire format.  Included by blsmi.hip.

// ---- hash to curve --------------------------------------------------------------------------------
// clear == 0: the hash point before its cofactor clearing (hash.cuh: swu_finish_g1; large g2pubs aggregates only)
KERNEL2 k_hash_g1(const u8* msgs, const u64* off, u8* out, size_t n, int clear, int* special) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    G1Aff h;
    hash_g1(h, msgs + off[tt], (size_t)(off[tt + 1] - off[tt]), clear, t < n ? special : nullptr);
    if (t < n) store_g1(out + 96 * t, h);
}
KERNEL k_hash_g2(const u8* msgs, const u64* off, u8* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    G2Aff h;
    hash_g2(h, msgs + off[tt], (size_t)(off[tt + 1] - off[tt]));
    if (t < n) store_g2(out + 192 * t, h);
}
KERNEL k_hash_g2_domain(const u8* msgs32, const u8* domain, u8* out, size_t n) {
    static_assert(WG == 64, "hash_g2_with_domain_wave: one wave per workgroup");
    __shared__ u32 lds[TAI_WAVE_LDS_WORDS];
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    G2Aff h;
    hash_g2_with_domain_wave(h, msgs32, (size_t)blockIdx.x * WG, n, domain, lds);   // the wave shares its messages' search
    if (t < n) store_g2(out + 192 * t, h);
}
// ---- small batches, the hash split in two: these kernels do SHA-256 and the SWU maps (or the try-and-increment search) --
// one Fq exponentiation chain per lane, which is what a lane is good at -- and leave the mapped points as wire-format field
// elements; the curve arithmetic that follows (isogeny, sum, cofactor clearing: wide and shallow) runs as a level program
// of the latency path (k_lat.hip: hashfin1 / hashfin2 / cofac2), one message per wave.  pts: 2 points per message.
KERNEL2 k_swu_g1_two_lanes(const u8* msgs, const u64* off, u8* pts, size_t n) {
    hash_prio();
    const size_t idx = (size_t)blockIdx.x * WG + threadIdx.x, t = idx >> 1, tt = t < n ? t : n - 1;
    u32 d[8];
    sha256_msg(d, 1, 0x01, msgs + off[tt], (size_t)(off[tt + 1] - off[tt]));
    G1Aff p;
    swu_g1_helper(p, hp_from_digest(d, (u32)(idx & 1)));
    if (t < n) store_g1(pts + 96 * idx, p);
}
// ... and the rest of HashG1 for MID-SIZE batches as a throughput kernel, one message per lane: the sum of the two mapped points, the
// 11-isogeny and the cofactor clearing (hash.go:306-321; hash.cuh: swu_finish_g1).  The one-kernel k_hash_g1 runs a message per lane from
// start to end -- 16 384 messages are 256 waves and take the 2.0 ms a full chip's 65 536 take; with the two square-root chains of a message
// on two lanes (k_swu_g1_two_lanes: 512 waves) and this tail behind them the same hash is 0.55 + 0.8 ms.
KERNEL2 k_hash_g1_finish(const u8* pts, u8* out, size_t n, int clear, int* special) {
    hash_prio();
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    const G1Aff p1 = load_g1(pts + 192 * tt), p2 = load_g1(pts + 192 * tt + 96);
    G1Aff h;
    swu_finish_g1(h, p1, p2, clear, t < n ? special : nullptr);
    if (t < n) store_g1(out + 96 * t, h);
}
// ... for the messages k_hash_g1_finish_quad (k_hash_quad.hip) flagged: the same tail with the reference's special cases, a message per lane
KERNEL2 k_hash_g1_finish_redo(const u8* pts, const u8* good, u8* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const bool mine = t < n && !good[t];
    if (!__any(mine)) return;
    const size_t tt = t < n ? t : n - 1;
    const G1Aff p1 = load_g1(pts + 192 * tt), p2 = load_g1(pts + 192 * tt + 96);
    G1Aff h;
    swu_finish_g1(h, p1, p2, 1, nullptr);
    if (mine) store_g1(out + 96 * t, h);
}
KERNEL k_swu_g2_two_lanes(const u8* msgs, const u64* off, u8* pts, size_t n) {
    hash_prio();
    const size_t idx = (size_t)blockIdx.x * WG + threadIdx.x, t = idx >> 1, tt = t < n ? t : n - 1;
    u32 d[8];
    sha256_msg(d, 1, 0x01, msgs + off[tt], (size_t)(off[tt + 1] - off[tt]));
    G2Aff p;
    swu_g2_helper(p, hp2_from_digest(d, (u32)(idx & 1)));
    if (t < n) store_g2(pts + 192 * idx, p);
}
// The same two kernels with one WAVE per map, for the smallest calls: every lane computes the same values and the map's square-root
// exponentiation -- a chain of 379 dependent squarings with nothing to run beside it -- runs with one limb per lane
// (fp_row.cuh: 219 us against 517 us for the chain).  Grid: 2 n workgroups of one wave; lane 0 stores.
__global__ void __launch_bounds__(64, 2) k_swu_g1_waves(const u8* msgs, const u64* off, u8* pts, size_t n) {
    const size_t idx = blockIdx.x, t = idx >> 1;
    u32 d[8];
    sha256_msg(d, 1, 0x01, msgs + off[t], (size_t)(off[t + 1] - off[t]));
    G1Aff p;
    swu_g1_helper_wave(p, hp_from_digest(d, (u32)(idx & 1)));
    if (threadIdx.x == 0) store_g1(pts + 96 * idx, p);
}
__global__ void __launch_bounds__(64) k_swu_g2_waves(const u8* msgs, const u64* off, u8* pts, size_t n) {
    const size_t idx = blockIdx.x, t = idx >> 1;
    u32 d[8];
    sha256_msg(d, 1, 0x01, msgs + off[t], (size_t)(off[t + 1] - off[t]));
    G2Aff p;
    swu_g2_helper_wave(p, hp2_from_digest(d, (u32)(idx & 1)));
    if (threadIdx.x == 0) store_g2(pts + 192 * idx, p);
}
// ... and with a ROW of sixteen lanes per map (four maps per wave) for calls in between (round 6): above the hand-over of the wave kernels a lane per map
// (k_swu_g?_two_lanes) takes 0.53 / 1.05 ms whatever the count -- n messages are n / 32 waves -- while 2 n rows are n / 2 waves: every SIMD has one
// from 2 048 messages.  Every lane of a row computes the same values; the exponentiations run with a limb per lane of the row (fp_row.cuh: fp_pow_row16).
__global__ void __launch_bounds__(64, 2) k_swu_g1_rows(const u8* msgs, const u64* off, u8* pts, size_t n) {
    hash_prio();
    const size_t idx = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 4, t = idx >> 1, tt = t < n ? t : n - 1;
    u32 d[8];
    sha256_msg(d, 1, 0x01, msgs + off[tt], (size_t)(off[tt + 1] - off[tt]));
    G1Aff p;
    swu_g1_helper_row16(p, hp_from_digest(d, (u32)(idx & 1)));
    if (t < n && !(threadIdx.x & 15)) store_g1(pts + 96 * idx, p);
}
__global__ void __launch_bounds__(64) k_swu_g2_rows(const u8* msgs, const u64* off, u8* pts, size_t n) {
    hash_prio();
    const size_t idx = ((size_t)blockIdx.x * 64 + threadIdx.x) >> 4, t = idx >> 1, tt = t < n ? t : n - 1;
    u32 d[8];
    sha256_msg(d, 1, 0x01, msgs + off[tt], (size_t)(off[tt + 1] - off[tt]));
    G2Aff p;
    swu_g2_helper_row16(p, hp2_from_digest(d, (u32)(idx & 1)));
    if (t < n && !(threadIdx.x & 15)) store_g2(pts + 192 * idx, p);
}
// the try-and-increment search of HashG2WithDomain with eight lanes per message (eight candidates per round)
KERNEL k_tai_g2_lanes8(const u8* msgs32, const u8* domain, u8* pts, size_t n) {
    const size_t idx = (size_t)blockIdx.x * WG + threadIdx.x, t = idx >> 3, tt = t < n ? t : n - 1;
    G2Aff p; p.inf = 0;
    tai_g2_group8(p.x, p.y, msgs32 + 32 * tt, domain, (int)(idx & 7));
    if (t < n && !(idx & 7)) store_g2(pts + 192 * t, p);
}
// ... and with eight WAVES per message for the smallest calls (hash.cuh: tai_g2_waves8): one 512-thread workgroup per message
__global__ void __launch_bounds__(512) k_tai_g2_waves8(const u8* msgs32, const u8* domain, u8* pts, size_t n) {
    __shared__ i32 lds[8 + 5 * NL];
    const size_t t = blockIdx.x;
    G2Aff p; p.inf = 0;
    tai_g2_waves8(p.x, p.y, msgs32 + 32 * t, domain, lds);
    if (threadIdx.x == 0) store_g2(pts + 192 * t, p);
}
// Messages whose level program met an exceptional step (good[t] == 0: equal / opposite mapped points, an isogeny pole, a
// result at infinity) are hashed again by the one-lane routines, which follow the reference's steps literally.  A wave
// without such a message -- every wave, in practice -- returns at once.
KERNEL2 k_hash_g1_redo(const u8* msgs, const u64* off, const u8* good, u8* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x, tt = t < n ? t : n - 1;
    const bool mine = t < n && !good[t];
    if (!__any(mine)) return;
    G1Aff h;
    hash_g1(h, msgs + off[tt], (size_t)(off[tt + 1] - off[tt]));
    if (mine) store_g1(out + 96 * t, h);
}
KERNEL k_hash_g2_redo(const u8* msgs, const u64* off, const u8* good, u8* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x, tt = t < n ? t : n - 1;
    const bool mine = t < n && !good[t];
    if (!__any(mine)) return;
    G2Aff h;
    hash_g2(h, msgs + off[tt], (size_t)(off[tt + 1] - off[tt]));
    if (mine) store_g2(out + 192 * t, h);
}
// the search and the root alone (the point before ScaleByCofactor, wire format); k_cofac2_pair (k_hash_pair.hip) finishes
KERNEL k_tai_g2_wave(const u8* msgs32, const u8* domain, u8* pts, size_t n) {
    __shared__ u32 lds[TAI_WAVE_LDS_WORDS];
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    G2Aff p;
    tai_g2_wave(p, msgs32, (size_t)blockIdx.x * WG, n, domain, lds);
    if (t < n) store_g2(pts + 192 * t, p);
}
KERNEL k_hash_g2_domain_redo(const u8* msgs32, const u8* domain, const u8* good, u8* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x, tt = t < n ? t : n - 1;
    const bool mine = t < n && !good[t];
    if (!__any(mine)) return;
    G2Aff h;
    hash_g2_with_domain(h, msgs32 + 32 * tt, domain);
    if (mine) store_g2(out + 192 * t, h);
}
KERNEL k_write_generators(u8* g1, u8* g2) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    G1Aff a; a.x = C_G1X; a.y = C_G1Y; a.inf = 0;
    G2Aff b; b.x = C_G2X; b.y = C_G2Y; b.inf = 0;
    store_g1(g1, a);
    store_g2(g2, b);
}
