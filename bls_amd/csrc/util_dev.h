// util_dev.h -- host-callable helpers defined in k_util.hip (kept out of kernels.h: that unit needs nothing of the field library).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <functional>
namespace blsmi_util {
// Duplicate-message rejection of VerifyAggregate (g2pubs/bls.go:245-261) over n messages resident on the device
// (message i = d_msgs[d_off[i] .. d_off[i+1]), d_off = n+1 uint64): *d_flag (a device int) is set nonzero when some
// message is empty or occurs twice (bit 0), or the table gave up and the caller must fall back to the sort on the host (bit 1).
// Work is enqueued on `s`; `scratch(bytes)` hands out device temporaries (at most 24 n bytes) that stay valid until `s` has run it.  Returns a hipError_t as int.
int dup_check_async(const void* d_msgs, const void* d_off, size_t n, uint64_t key0, uint64_t key1, int* d_flag, hipStream_t s,
                    const std::function<void*(size_t)>& scratch);
// Grouping the MSM's (bucket, point) items (blsmi.hip: msm_bucket_glv_dev): n pairs of (uint32 key, uint32 value), the low `bits` bits of the
// key significant, sorted by key with the device radix sort of rocPRIM (through hipCUB) on `s`.  k[0] / v[0] hold the input; the
// sorted pairs end up in *k_sorted / *v_sorted (one of the two buffers of each kind; the other is scratch).  `scratch` as above.
int sort_pairs_async(uint32_t* k[2], uint32_t* v[2], size_t n, int bits, hipStream_t s, const std::function<void*(size_t)>& scratch,
                     uint32_t** k_sorted, uint32_t** v_sorted);
}
