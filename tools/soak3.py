"""Round-3 soak of the one-element-per-wave kernels (fp_row.cuh): hash-to-curve of both groups, the try-and-increment search and both
decompressions at random small batch sizes against the one-element-per-lane kernels (latency path off), and samples against the oracle.
python tools/soak3.py [seconds]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from bls_amd import engine
from oracle import refcpu as RC
engine.init(0)
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "31337")))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
t0 = time.time(); rounds = 0; checked = 0
def both(fn, *args):
    engine.set_latency_threshold(8192); a = fn(*args)
    engine.set_latency_threshold(0); b = fn(*args)
    engine.set_latency_threshold(8192)
    return a, b
while time.time() - t0 < budget:
    n = int(rng.integers(1, 513))
    msgs = [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in rng.integers(0, 120, size=n)]
    for fn, ref in ((engine.hash_g1_batch, RC.hash_g1), (engine.hash_g2_batch, RC.hash_g2)):
        a, b = both(fn, msgs)
        assert np.array_equal(a, b), ("hash", n)
        i = int(rng.integers(0, n)); assert a[i].tobytes() == ref(msgs[i]); checked += 1
    m = int(rng.integers(1, 129))
    m32 = [hashlib.sha256(bytes(rng.integers(0, 256, size=8, dtype=np.uint8))).digest() for _ in range(m)]
    dom = bytes(rng.integers(0, 256, size=8, dtype=np.uint8))
    a, b = both(engine.hash_g2_with_domain_batch, m32, dom)
    assert np.array_equal(a, b), ("domain", m)
    i = int(rng.integers(0, m)); assert a[i].tobytes() == RC.hash_g2_with_domain(m32[i], dom); checked += 1
    # decompression: valid points, random junk, flipped bits
    k = int(rng.integers(1, 513))
    sk = rng.integers(0, 256, size=(k, 32), dtype=np.uint8); sk[:, 0] &= 0x3f
    p1, _ = engine.g1_mul_generator_batch(sk.reshape(-1), k); p2, _ = engine.g2_mul_generator_batch(sk.reshape(-1), k)
    c1 = engine.g1_compress_batch(p1.reshape(-1), k).copy(); c2 = engine.g2_compress_batch(p2.reshape(-1), k).copy()
    for c in (c1, c2):
        for j in rng.integers(0, k, size=max(1, k // 7)):
            mode = int(rng.integers(0, 4))
            if mode == 0: c[j] = rng.integers(0, 256, size=c.shape[1], dtype=np.uint8)
            elif mode == 1: c[j, int(rng.integers(0, c.shape[1]))] ^= 1 << int(rng.integers(0, 8))
            elif mode == 2: c[j] = 0; c[j, 0] = 0xc0
            else: c[j, 0] ^= 0x20
    for fn, c in ((engine.g1_decompress_batch, c1), (engine.g2_decompress_batch, c2)):
        for chk in (True, False):
            a, b = both(fn, c.reshape(-1), k, chk)
            for x, y in zip(a, b):
                assert np.array_equal(np.asarray(x), np.asarray(y)), ("decompress", k, chk)
    rounds += 1
print("soak3 ok: %d rounds, %d oracle samples, %.0f s" % (rounds, checked, time.time() - t0))
