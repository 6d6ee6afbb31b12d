// k_util.hip -- byte/index utilities of the verify path that are not field arithmetic: the duplicate-message rejection of
// VerifyAggregate (g2pubs/bls.go:245-261, g1pubs/bls.go:257-273) on the device, for callers whose messages are resident
// in HBM (the *_dev entry points) and for large host batches whose messages have been copied there anyway.
//
// The reference sorts the messages bytewise and compares neighbours; lastMsg starts as nil and bytes.Equal(m, nil) holds
// for an empty m, so an empty message is rejected as well.  The verdict "some message occurs twice, or one is empty" is
// reproduced with an open-addressing table, as on the host (verify_host.inc has_duplicates): a keyed 64-bit fingerprint per
// message picks the slot, a message claims the first free slot of its probe sequence with an atomic compare-and-swap, and on
// its way it compares itself -- fingerprint first, then byte for byte -- with every message already sitting there.  Of two
// equal messages the later claimant walks the earlier one's probe sequence and meets it.  Fingerprints only choose what gets
// compared; the verdict rests on exact comparisons.  Messages are attacker-supplied, so the fingerprint is keyed per
// process (the caller passes the key); a probe sequence beyond PROBE_LIMIT (load factor <= 1/2: astronomically unlikely with
// an unknown key) raises flag bit 1 and the caller falls back to the reference's sort on the host.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include "util_dev.h"

namespace {
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
constexpr int PROBE_LIMIT = 512;

__device__ __forceinline__ u64 mix(u64 h, u64 w, u64 k1, u64 c) { h ^= w; h *= k1; h ^= h >> 32; h *= c; h ^= h >> 29; return h; }

// one lane per message: its fingerprint; *flag bit 0 raised for an empty message
__global__ void __launch_bounds__(256) k_msg_fingerprint(const u8* msgs, const u64* off, size_t n, u64 k0, u64 k1, u64* fp, int* flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u8* m = msgs + off[i];
    const size_t len = (size_t)(off[i + 1] - off[i]);
    if (len == 0) atomicOr(flag, 1);
    u64 h = k0 ^ ((u64)len * k1);
    size_t k = 0;
    for (; k + 8 <= len; k += 8) {
        u64 w = 0;
        for (int b = 0; b < 8; b++) w |= (u64)m[k + b] << (8 * b);      // unaligned-safe little-endian load
        h = mix(h, w, k1, 0xff51afd7ed558ccdull);
    }
    if (k < len) {
        u64 w = 0;
        for (int b = 0; k + b < len; b++) w |= (u64)m[k + b] << (8 * b);
        h = mix(h, w, k1, 0xc4ceb9fe1a85ec53ull);
    }
    h *= 0x9e3779b97f4a7c15ull; h ^= h >> 31;
    fp[i] = h;
}
__device__ bool same_message(const u8* msgs, const u64* off, u32 a, u32 b) {
    const size_t la = (size_t)(off[a + 1] - off[a]), lb = (size_t)(off[b + 1] - off[b]);
    if (la != lb) return false;
    const u8* x = msgs + off[a]; const u8* y = msgs + off[b];
    for (size_t k = 0; k < la; k++) if (x[k] != y[k]) return false;
    return true;
}
// table: cap (a power of two >= 2 n) words, 0 = free, else message index + 1
__global__ void __launch_bounds__(256) k_dup_insert(const u64* fp, const u8* msgs, const u64* off, size_t n, u32* table, u32 mask, int* flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 mine = fp[i];
    u32 slot = (u32)mine & mask;
    for (int probes = 0;; probes++) {
        if (probes > PROBE_LIMIT) { atomicOr(flag, 2); return; }
        const u32 old = atomicCAS(&table[slot], 0u, (u32)i + 1u);
        if (old == 0) return;                                              // claimed a free slot: nobody equal was on the way
        const u32 j = old - 1;
        if (fp[j] == mine && same_message(msgs, off, (u32)i, j)) { atomicOr(flag, 1); return; }
        slot = (slot + 1) & mask;
    }
}
}  // namespace

namespace blsmi_util {
// *d_flag (device int) bit 0: a message is empty or occurs twice; bit 1: the table gave up (caller: fall back to the sort).
// Enqueues on s; temporaries (8 n bytes + the table, at most 16 n bytes) come from the caller's allocator and must live until s
// has run them.  Returns a hipError_t as int.
int dup_check_async(const void* d_msgs, const void* d_off, size_t n, uint64_t key0, uint64_t key1, int* d_flag, hipStream_t s,
                    const std::function<void*(size_t)>& scratch) {
    hipError_t e = hipMemsetAsync(d_flag, 0, sizeof(int), s);
    if (e != hipSuccess || n == 0) return (int)e;
    if (n >= 0x7fffffffull) { const int one = 1; return (int)hipMemcpyAsync(d_flag, &one, sizeof one, hipMemcpyHostToDevice, s); }   // not representable: reject
    size_t cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    u64* fp = (u64*)scratch(8 * n);
    u32* table = (u32*)scratch(4 * cap);
    if (!fp || !table) return (int)hipErrorOutOfMemory;
    if ((e = hipMemsetAsync(table, 0, 4 * cap, s)) != hipSuccess) return (int)e;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_msg_fingerprint, dim3(blocks), dim3(256), 0, s, (const u8*)d_msgs, (const u64*)d_off, n, (u64)key0, (u64)key1, fp, d_flag);
    hipLaunchKernelGGL(k_dup_insert, dim3(blocks), dim3(256), 0, s, (const u64*)fp, (const u8*)d_msgs, (const u64*)d_off, n, table, (u32)(cap - 1), d_flag);
    return (int)hipGetLastError();
}
// (key, value) pairs sorted by the low `bits` key bits: rocPRIM's device radix sort.  2^24 pairs, 19 bits: 0.44 ms on an MI355X, where
// the one-pass atomic scatter this replaced in the MSM (count + claim a slot + scattered 4-byte store per item) took 0.99.
int sort_pairs_async(uint32_t* k[2], uint32_t* v[2], size_t n, int bits, hipStream_t s, const std::function<void*(size_t)>& scratch,
                     uint32_t** k_sorted, uint32_t** v_sorted) {
    if (n >= 0x7fffffffull) return (int)hipErrorInvalidValue;
    hipcub::DoubleBuffer<uint32_t> dk(k[0], k[1]), dv(v[0], v[1]);
    size_t bytes = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, dk, dv, (int)n, 0, bits, s);
    if (e != hipSuccess) return (int)e;
    void* tmp = scratch(bytes ? bytes : 1);
    if (!tmp) return (int)hipErrorOutOfMemory;
    e = hipcub::DeviceRadixSort::SortPairs(tmp, bytes, dk, dv, (int)n, 0, bits, s);
    *k_sorted = dk.Current(); *v_sorted = dv.Current();
    return (int)e;
}
}  // namespace blsmi_util
