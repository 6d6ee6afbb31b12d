"""One-off soak (GPU box): random tuples through both paths, outputs compared with each other and sampled against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from bls_amd import engine
from oracle import refcpu as RC, pyref as P
engine.init(0)
rng = np.random.default_rng(2024)
def scal(n):
    raw = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); raw[:, 0] &= 0x3f; raw[:, 31] |= 1
    return raw
n = 16384
g1, _ = engine.g1_mul_batch(RC.g1_generator() * n, scal(n).reshape(-1), n)
g2, _ = engine.g2_mul_batch(RC.g2_generator() * n, scal(n).reshape(-1), n)
outs = {}
for thr in (0, 4096):
    engine.set_latency_threshold(thr)
    parts = [engine.pairing_batch(g1[i:i + 4096].reshape(-1), g2[i:i + 4096].reshape(-1), 4096) for i in range(0, n, 4096)]
    outs[thr] = np.concatenate(parts)
assert np.array_equal(outs[0], outs[4096]), "paths differ"
idx = rng.integers(0, n, size=200)
t0 = time.time()
for i in idx:
    assert np.array_equal(outs[0][i], RC.pairing_batch(g1[i].tobytes(), g2[i].tobytes(), 1)[0]), i
print("pairing soak ok: %d tuples on both paths, %d oracle samples (%.1f s)" % (n, len(idx), time.time() - t0))
# hashes, both packages, odd lengths, both the one-lane and the two-lane kernels
msgs = [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in rng.integers(0, 200, size=3000)]
for name, fn, ref in (("g1", engine.hash_g1_batch, RC.hash_g1), ("g2", engine.hash_g2_batch, RC.hash_g2)):
    engine.set_latency_threshold(8192); a = fn(msgs[:2000])
    engine.set_latency_threshold(0); b = fn(msgs[:2000])
    assert np.array_equal(a, b), name
    for i in rng.integers(0, 2000, size=60):
        assert a[i].tobytes() == ref(msgs[i]), (name, i)
    engine.set_latency_threshold(8192)
dom = bytes(rng.integers(0, 256, size=8, dtype=np.uint8))
m32 = [bytes(rng.integers(0, 256, size=32, dtype=np.uint8)) for _ in range(600)]
a = engine.hash_g2_with_domain_batch(m32, dom); engine.set_latency_threshold(0); b = engine.hash_g2_with_domain_batch(m32, dom); engine.set_latency_threshold(8192)
assert np.array_equal(a, b)
for i in rng.integers(0, 600, size=25):
    assert a[i].tobytes() == RC.hash_g2_with_domain(m32[i], dom), i
print("hash soak ok")
# scalar multiplication (level program with SEL levels against the windowed kernels), full-range and short scalars
m = 3000
k = scal(m); k[::7, :20] = 0; k[::11] = 0
for name, mul, pts, ref in (("g1", engine.g1_mul_batch, g1[:m], RC.g1_mul), ("g2", engine.g2_mul_batch, g2[:m], RC.g2_mul)):
    engine.set_latency_threshold(8192); a, ia = mul(pts.reshape(-1), k.reshape(-1), m)
    engine.set_latency_threshold(0); b, ib = mul(pts.reshape(-1), k.reshape(-1), m)
    engine.set_latency_threshold(8192)
    assert np.array_equal(a, b) and np.array_equal(ia, ib), name
    assert ia[::11].all() and not ia[1::11][:50].any()
    for i in rng.integers(0, m, size=40):
        e = ref(pts[i].tobytes(), k[i].tobytes())
        assert (e is None and ia[i]) or a[i].tobytes() == e, (name, i)
print("scalar multiplication soak ok")
# compressed wire format: valid points through both decompression paths (subgroup test as a level program / in the kernel)
for name, comp, dec, pts, cb in (("g1", engine.g1_compress_batch, engine.g1_decompress_batch, g1[:m], 48), ("g2", engine.g2_compress_batch, engine.g2_decompress_batch, g2[:m], 96)):
    c = comp(pts.reshape(-1), m)
    engine.set_latency_threshold(8192); ra = dec(np.asarray(c).reshape(-1), m, True)
    engine.set_latency_threshold(0); rb = dec(np.asarray(c).reshape(-1), m, True)
    engine.set_latency_threshold(8192)
    for x, y in zip(ra, rb):
        assert np.array_equal(np.asarray(x), np.asarray(y)), name
    assert np.array_equal(np.asarray(ra[0]).reshape(m, -1), pts.reshape(m, -1)), name
print("decompression soak ok")
# differential fuzz: random sizes, every batch entry point on both paths (latency programs vs throughput kernels)
import random
rnd = random.Random(7)
def both(fn):
    engine.set_latency_threshold(8192); a = fn()
    engine.set_latency_threshold(0); b = fn()
    engine.set_latency_threshold(8192)
    return a, b
def same(a, b):
    if isinstance(a, tuple):
        return all(same(x, y) for x, y in zip(a, b))
    if a is None or b is None or isinstance(a, (bytes, bool)):
        return a == b
    return np.array_equal(np.asarray(a), np.asarray(b))
for it in range(40):
    nn = rnd.choice([1, 2, 3, 5, 17, 63, 64, 65, 129, 300])
    i0 = rnd.randrange(0, n - nn)
    p1, p2 = g1[i0:i0 + nn], g2[i0:i0 + nn]
    kk = scal(nn)
    flags = np.array([rnd.random() < 0.1 for _ in range(nn)], dtype=np.uint8)
    ops = {
        "g1_sum": lambda: engine.g1_sum(p1.reshape(-1), nn, flags), "g2_sum": lambda: engine.g2_sum(p2.reshape(-1), nn, flags),
        "g1_msm": lambda: engine.g1_msm(p1.reshape(-1), kk.reshape(-1), nn), "g2_msm": lambda: engine.g2_msm(p2.reshape(-1), kk.reshape(-1), nn),
        "miller": lambda: engine.miller_loop_batch(p1.reshape(-1), p2.reshape(-1), nn),
        "pairing": lambda: engine.pairing_batch(p1.reshape(-1), p2.reshape(-1), nn),
        "fe": lambda: engine.final_exponentiation_batch(engine.miller_loop_batch(p1.reshape(-1), p2.reshape(-1), nn)),
        "fq12_product": lambda: engine.fq12_product(engine.miller_loop_batch(p1.reshape(-1), p2.reshape(-1), nn)),
    }
    name = rnd.choice(list(ops))
    a, b = both(ops[name])
    assert same(a, b), (name, nn, it)
print("differential fuzz ok")
