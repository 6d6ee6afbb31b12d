"""Multi-GPU sharding of batch verification (BASELINE config 4): one process per GPU, contiguous block
partition of the n independent tuples, no collective in the data path, and ONE small collective at the
end -- an all-reduce (bitwise OR, realised as SUM over disjoint shards) of the zero-padded pass/fail
bitmap, ceil(n/8) bytes, over RCCL/xGMI (backend "nccl" on ROCm).  A single n-way VerifyAggregate
additionally all-gathers the per-shard Fq12 Miller-loop partial products (world x 576 B) -- see
DESIGN.md (e).

The per-shard verifier is injected (`verify_shard`), which keeps this module testable on CPU with the
gloo backend and world_size 2 (tests/test_dist_cpu.py); in production it is bls_amd.engine's batch call."""
import numpy as np


def shard_bounds(n, rank, world):
    """Contiguous block partition i -> rank floor(world*i/n): [lo, hi) of this rank."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def sharded_verify_bitmap(n, verify_shard, rank, world, all_reduce_sum_u8):
    """Each rank verifies its block and contributes its bits; returns the full LSB-first bitmap on every
    rank.  `verify_shard(lo, hi)` -> array of hi-lo booleans.  `all_reduce_sum_u8(np.uint8 array)`
    performs an in-place SUM all-reduce (disjoint bit ownership makes SUM == OR as long as shard borders
    do not split a byte between ranks; border bytes are handled by widening to uint16 lanes)."""
    lo, hi = shard_bounds(n, rank, world)
    ok = np.asarray(verify_shard(lo, hi), dtype=bool)
    assert ok.shape == (hi - lo,)
    bits = np.zeros(n, dtype=np.uint8)
    bits[lo:hi] = ok
    packed = np.packbits(bits, bitorder="little")          # ceil(n/8) bytes, this rank's bits only
    wide = packed.astype(np.int32)                          # SUM of disjoint bit sets never carries across bytes
    wide = all_reduce_sum_u8(wide)
    return wide.astype(np.uint8)


def torch_all_reduce(device=None):
    """all-reduce closure over torch.distributed (RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist

    def fn(arr):
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()
    return fn


def unpack_bitmap(bitmap, n):
    return np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")[:n].astype(bool)
