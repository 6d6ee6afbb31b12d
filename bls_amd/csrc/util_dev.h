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
}
