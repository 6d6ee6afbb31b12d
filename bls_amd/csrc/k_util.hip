// k_util.hip -- byte/index utilities of the verify path that are not field arithmetic: the duplicate-message rejection of
// VerifyAggregate (g2pubs/bls.go:245-261, g1pubs/bls.go:257-273) on the device, for callers whose messages are resident
// in HBM (the *_dev entry points) and for large host batches whose messages have been copied there anyway.
//
// The reference sorts the messages bytewise and compares neighbours; lastMsg starts as nil and bytes.Equal(m, nil) holds
// for an empty m, so an empty message is rejected as well.  The verdict "some message occurs twice, or one is empty" is
// reproduced as: keyed 64-bit fingerprint per message -> radix sort of (fingerprint, index) (rocPRIM through hipCUB) ->
// every message compares itself byte for byte with the earlier members of its run of equal fingerprints.  Fingerprints
// only choose what gets compared; the verdict rests on exact comparisons.  Messages are attacker-supplied, so the
// fingerprint is keyed per process (the caller passes the key): without the key runs have length 1 (+ true duplicates).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include "util_dev.h"

namespace {
typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;

__device__ __forceinline__ u64 mix(u64 h, u64 w, u64 k1, u64 c) { h ^= w; h *= k1; h ^= h >> 32; h *= c; h ^= h >> 29; return h; }

// one lane per message: fp[i], idx[i] = i; *flag raised for an empty message
__global__ void __launch_bounds__(256) k_msg_fingerprint(const u8* msgs, const u64* off, size_t n, u64 k0, u64 k1, u64* fp, u32* idx, int* flag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u8* m = msgs + off[i];
    const size_t len = (size_t)(off[i + 1] - off[i]);
    if (len == 0) *flag = 1;
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
    fp[i] = h; idx[i] = (u32)i;
}
__device__ bool same_message(const u8* msgs, const u64* off, u32 a, u32 b) {
    const size_t la = (size_t)(off[a + 1] - off[a]), lb = (size_t)(off[b + 1] - off[b]);
    if (la != lb) return false;
    const u8* x = msgs + off[a]; const u8* y = msgs + off[b];
    for (size_t k = 0; k < la; k++) if (x[k] != y[k]) return false;
    return true;
}
// sorted position t compares its message with the earlier members of its fingerprint run
__global__ void __launch_bounds__(256) k_dup_in_runs(const u64* fp, const u32* idx, const u8* msgs, const u64* off, size_t n, int* flag) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t == 0 || t >= n) return;
    const u64 mine = fp[t];
    for (size_t j = t; j-- > 0 && fp[j] == mine;) {
        if (same_message(msgs, off, idx[t], idx[j])) { *flag = 1; return; }
        if (*(volatile int*)flag) return;                                  // someone already found a duplicate
    }
}
}  // namespace

namespace blsmi_util {
// *d_flag (device int, cleared by the caller or here) becomes nonzero when a message is empty or occurs twice.
// Enqueues on s; temporaries (20 n bytes + the sort's scratch) come from the caller's allocator and must live until s has run them.
// Returns a hipError_t as int.
int dup_check_async(const void* d_msgs, const void* d_off, size_t n, uint64_t key0, uint64_t key1, int* d_flag, hipStream_t s,
                    const std::function<void*(size_t)>& scratch) {
    hipError_t e = hipMemsetAsync(d_flag, 0, sizeof(int), s);
    if (e != hipSuccess || n == 0) return (int)e;
    if (n >= 0x7fffffffull) { const int one = 1; return (int)hipMemcpyAsync(d_flag, &one, sizeof one, hipMemcpyHostToDevice, s); }   // not representable: reject
    size_t tmp_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const u64*)nullptr, (u64*)nullptr, (const u32*)nullptr, (u32*)nullptr, (int)n, 0, 64, s);
    u64* fp0 = (u64*)scratch(8 * n); u64* fp1 = (u64*)scratch(8 * n);
    u32* ix0 = (u32*)scratch(4 * n); u32* ix1 = (u32*)scratch(4 * n);
    void* tmp = scratch(tmp_bytes ? tmp_bytes : 1);
    if (!fp0 || !fp1 || !ix0 || !ix1 || !tmp) return (int)hipErrorOutOfMemory;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_msg_fingerprint, dim3(blocks), dim3(256), 0, s, (const u8*)d_msgs, (const u64*)d_off, n, (u64)key0, (u64)key1, fp0, ix0, d_flag);
    if ((e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, fp0, fp1, ix0, ix1, (int)n, 0, 64, s)) != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_dup_in_runs, dim3(blocks), dim3(256), 0, s, (const u64*)fp1, (const u32*)ix1, (const u8*)d_msgs, (const u64*)d_off, n, d_flag);
    return (int)hipGetLastError();
}
}  // namespace blsmi_util
