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


def sharded_verify_aggregate(group, shard_msgs, shard_pks, sig, rank, world, all_gather_bytes, engine=None):
    """One n-way VerifyAggregate (g2pubs/bls.go:240-270) whose (message, public key) pairs are block-sharded
    over `world` ranks.  Per rank: the shard's Miller-loop product on its GPU (started first, on its own host thread), and
    beside it the duplicate screening -- an all-gather of 8-byte message fingerprints, of which rank r sorts the share with
    value = r mod world; full 33-byte keys are exchanged and compared exactly only when some rank found a repeat.  Then one
    all-gather of the 576-byte Fq12 partials, after which every rank multiplies the partials, runs the signature-side
    Miller loop and one final exponentiation, and compares.  Returns the same boolean on every rank.
    `all_gather_bytes(b: bytes) -> list[bytes]` is the collective (RCCL via torch.distributed in production, gloo in the
    CPU test); `engine` defaults to bls_amd.engine.

    Screens, agreed on by all ranks before any of them can return (a one-byte status rides on the first
    all-gather): a shard whose key and message counts differ (g2pubs/bls.go:241-243), an empty message, a key or
    the signature at infinity (all-zero record; the reference panics in MillerLoop there) -> False everywhere.
    Duplicate rejection is exact (message_keys: messages of up to 32 bytes verbatim, longer ones by SHA-256 digest --
    equal messages always collide, distinct ones only with a SHA-256 collision)."""
    if engine is None:
        from . import engine as _e
        engine = _e
    pk_bytes = 192 if group == "g2pubs" else 96
    pk_raw = bytes(shard_pks)
    status = 0
    if len(pk_raw) != pk_bytes * len(shard_msgs):
        status |= 2                                            # len(pubKeys) != len(msgs)
    if (hasattr(shard_msgs, "off") and (np.diff(shard_msgs.off.astype(np.int64)) == 0).any()) or \
            (not hasattr(shard_msgs, "off") and any(len(m) == 0 for m in shard_msgs)):
        status |= 1
    if not any(bytes(sig)):
        status |= 4                                            # signature at infinity
    # The shard's Miller-loop product does not depend on the screening: it runs on the GPU (its own host thread; the C call
    # releases the interpreter lock) while this thread exchanges the keys and looks for duplicates.
    fut = _executor().submit(engine.aggregate_partial, group, shard_msgs, pk_raw) if status == 0 else None
    failure = None
    part, bad = np.zeros(72, dtype=np.uint64), False
    try:
        # Duplicate rejection (g2pubs/bls.go:245-261) across ranks.  Equal messages have equal 33-byte keys, hence equal 64-bit
        # fingerprints: the ranks exchange the FINGERPRINTS (8 bytes per message), rank r looks for a repeated value among those
        # with fingerprint = r mod world -- 1/world of the sorting each -- and only when some rank finds one are the full keys
        # exchanged and compared exactly (a true duplicate, or a 2^-64 accident).  The fingerprint is KEYED with a nonce the
        # ranks agreed on (messages are attacker-supplied: an unkeyed one could be driven into collisions at will, which
        # costs the full-key exchange on every call -- never a wrong verdict).
        fkey = _fingerprint_key(all_gather_bytes, world)
        keys = np.frombuffer(message_keys(shard_msgs), dtype=np.uint8).reshape(-1, 33)
        fp = row_fingerprints(keys, fkey) if keys.shape[0] else np.zeros(0, np.uint64)
        gathered = _tagged(all_gather_bytes, b"F", fp.tobytes() + bytes([status]))
        any_status = 0
        for g in gathered:
            any_status |= g[-1]
        if any_status:                                         # empty message, bad shard, infinity: reject on every rank
            return False
        allfp = np.concatenate([np.frombuffer(g[:-1], dtype=np.uint64) for g in gathered]) if gathered else np.zeros(0, np.uint64)
        mine = np.sort(allfp[allfp % np.uint64(world) == np.uint64(rank)])
        suspect = bool(mine.size > 1 and (mine[1:] == mine[:-1]).any())
        flags = _tagged(all_gather_bytes, b"S", bytes([1 if suspect else 0]), 1)
        if any(f[0] for f in flags):
            gk = _tagged(all_gather_bytes, b"X", keys.tobytes())
            allk = np.concatenate([np.frombuffer(g, dtype=np.uint8).reshape(-1, 33) for g in gk])
            if has_duplicate_rows(allk):
                return False                                   # some message occurs twice
        try:
            part, bad = fut.result()
        except Exception as e:  # noqa: BLE001 -- a HIP error / out of memory on THIS rank: the peers are about to enter the
            failure = e         # all-gather of the partials and would wait there forever; tell them through it, then raise
    finally:
        if fut is not None:
            import concurrent.futures
            concurrent.futures.wait([fut])                     # never leave the shard's call running behind a return
    parts = _tagged(all_gather_bytes, b"P", np.asarray(part, dtype=np.uint64).tobytes() + bytes([(1 if bad else 0) | (2 if failure is not None else 0)]), 577)
    if failure is not None:
        raise failure
    if any(p[-1] & 2 for p in parts):
        raise RuntimeError("sharded_verify_aggregate: the shard of rank(s) %s failed on its device" % [i for i, p in enumerate(parts) if p[-1] & 2])
    if any(p[-1] & 1 for p in parts):                              # a key at infinity on some rank
        return False
    rhs = engine.fq12_product(np.frombuffer(b"".join(p[:-1] for p in parts), dtype=np.uint64))
    if group == "g2pubs":
        lhs = engine.miller_loop_batch(sig, engine_generator(engine, 2), 1)[0]
    else:
        lhs = engine.miller_loop_batch(engine_generator(engine, 1), sig, 1)[0]
    fe = engine.final_exponentiation_batch(np.stack([lhs, rhs]))
    return bool(np.array_equal(fe[0], fe[1]))


_EXEC = None


def _executor():
    """a small pool per process for the shards' device calls (not one pool per VerifyAggregate; several workers so that
    VerifyAggregates issued from different threads do not queue behind one another -- the library leases them separate contexts)"""
    global _EXEC
    if _EXEC is None:
        import concurrent.futures
        _EXEC = concurrent.futures.ThreadPoolExecutor(max_workers=4, thread_name_prefix="blsmi-shard")
    return _EXEC


_FP_TEST_MASK = None      # test hook: AND every fingerprint with this mask (forces collisions between distinct messages)


class OutOfStep(RuntimeError):
    """the ranks' collective sequences have drifted apart (a restarted rank, a different group): a payload carried the wrong tag"""


def _tagged(all_gather_bytes, tag, payload, fixed_len=None):
    """all-gather of tag + payload; every part must carry the same tag (and length, when fixed): a rank that is one collective
    ahead or behind -- restarted into an existing group, or raised before its first call -- is detected instead of misparsed"""
    parts = all_gather_bytes(tag + payload)
    for i, p in enumerate(parts):
        if p[:1] != tag or (fixed_len is not None and len(p) != 1 + fixed_len):
            raise OutOfStep("sharded_verify_aggregate: rank %d sent a %r-tagged payload of %d bytes where %r (%s bytes) was expected"
                            % (i, bytes(p[:1]), len(p), tag, "any" if fixed_len is None else 1 + fixed_len))
    return [p[1:] for p in parts]


def _fingerprint_key(all_gather_bytes, world):
    """(k0, k1) of the keyed message fingerprint for THIS call, the same on every rank and unknown to whoever supplies the
    messages: every rank contributes 16 fresh random bytes through one tagged all-gather, the key is the XOR.  Derived per
    call (17 bytes per rank: nothing beside the fingerprints) -- a key cached per process made the first call a hidden collective
    that a restarted rank, or one joining a second group of the same size, would replay out of step."""
    import os
    parts = _tagged(all_gather_bytes, b"K", os.urandom(16), 16)
    acc = np.zeros(2, dtype=np.uint64)
    for p in parts:
        acc ^= np.frombuffer(p[:16], dtype=np.uint64)
    return (np.uint64(acc[0]), np.uint64(acc[1]) | np.uint64(1))


def row_fingerprints(keys, fkey=(np.uint64(0x9e3779b97f4a7c15), np.uint64(0xff51afd7ed558ccd))):
    """64-bit fingerprint of every row of an (n, 33) uint8 key array: equal rows share it.  Multiply-xorshift over the
    length byte and the four 8-byte words, keyed by fkey = (k0, odd k1)."""
    n = keys.shape[0]
    w = np.ascontiguousarray(keys[:, 1:33]).view(np.uint64).reshape(n, 4)
    k0, k1 = np.uint64(fkey[0]), np.uint64(fkey[1])
    with np.errstate(over="ignore"):
        h = (keys[:, 0].astype(np.uint64) + k0) * k1
        for j in range(4):
            h = (h ^ w[:, j]) * k1
            h ^= h >> np.uint64(29)
            h *= np.uint64(0xc4ceb9fe1a85ec53)
            h ^= h >> np.uint64(32)
    if _FP_TEST_MASK is not None:
        h = h & np.uint64(_FP_TEST_MASK)
    return h


def has_duplicate_rows(keys):
    """exact duplicate test over the rows of an (n, 33) uint8 array: a 64-bit fingerprint (xor of the four 8-byte words)
    is sorted first -- equal rows share it -- and only rows whose fingerprints collide are compared in full."""
    n = keys.shape[0]
    if n < 2:
        return False
    fp = row_fingerprints(keys)
    fs = np.sort(fp)                                            # the common case -- no two rows share a fingerprint -- needs no permutation
    if not (fs[1:] == fs[:-1]).any():
        return False
    order = np.argsort(fp, kind="stable")
    fps = fp[order]
    hit = fps[1:] == fps[:-1]
    idx = np.nonzero(np.concatenate([[False], hit]) | np.concatenate([hit, [False]]))[0]     # every member of a collision group
    rows = keys[order[idx]]
    rows = rows[np.lexsort(rows.T[::-1])]
    return bool((rows[1:] == rows[:-1]).all(axis=1).any())


def message_keys(msgs):
    """33-byte duplicate-detection key per message: messages of at most 32 bytes travel verbatim (length byte + bytes,
    zero padded: exact comparison), longer ones as 0xff + SHA-256 (equal messages always collide, distinct ones only with
    a SHA-256 collision).  Uniform 32-byte messages (the Eth2-era shape, BASELINE configs[3]) need no hashing at all."""
    import hashlib
    n = len(msgs)
    if hasattr(msgs, "buf") and hasattr(msgs, "off"):          # engine.PackedMsgs
        ln = np.diff(msgs.off.astype(np.int64))
        if n and (ln == 32).all():
            k = np.full((n, 33), 32, dtype=np.uint8)
            k[:, 1:] = msgs.buf[:32 * n].reshape(n, 32)
            return k.tobytes()
        msgs = [bytes(msgs.buf[int(msgs.off[i]):int(msgs.off[i + 1])]) for i in range(n)]
    if n and all(len(m) == 32 for m in msgs):
        k = np.full((n, 33), 32, dtype=np.uint8)
        k[:, 1:] = np.frombuffer(b"".join(bytes(m) for m in msgs), dtype=np.uint8).reshape(n, 32)
        return k.tobytes()
    out = bytearray()
    for m in msgs:
        m = bytes(m)
        out += (bytes([len(m)]) + m.ljust(32, b"\0")) if len(m) <= 32 else (b"\xff" + hashlib.sha256(m).digest())
    return bytes(out)


_G1_GEN = bytes.fromhex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
                        "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")
_G2_GEN = bytes.fromhex("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
                        "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                        "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"
                        "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")


def engine_generator(engine, group):
    """G1 / G2 generators in the affine wire format (g1.go:25-26, g2.go:26-29)."""
    return _G1_GEN if group == 1 else _G2_GEN


def torch_all_gather_bytes(device=None):
    """all-gather of variable-length byte strings over torch.distributed (lengths first, then padded payloads)."""
    import torch
    import torch.distributed as dist

    def fn(b):
        world = dist.get_world_size()
        ln = torch.tensor([len(b)], dtype=torch.int64, device=device)
        lens = [torch.zeros_like(ln) for _ in range(world)]
        dist.all_gather(lens, ln)
        mx = int(max(int(x.item()) for x in lens))
        buf = torch.zeros(mx, dtype=torch.uint8, device=device)
        if len(b):
            buf[:len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(buf.device)
        outs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(outs, buf)
        return [bytes(o.cpu().numpy()[:int(l.item())]) for o, l in zip(outs, lens)]
    return fn
