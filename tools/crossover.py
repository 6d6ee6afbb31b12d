"""Where the latency path (one tuple per wave) hands over to the throughput kernels: pairing_batch and g2pubs verify_batch
timed at the same n on both paths (tools/crossover.py, run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from bls_amd import engine as E

E.init(0)
g1, g2 = bench._gens()
rng = np.random.default_rng(3)
N = 32768
k = rng.integers(0, 256, size=(N, 32), dtype=np.uint8); k[:, 0] &= 0x3f
p1, _ = E.g1_mul_batch(g1 * N, k.reshape(-1), N)
p2, _ = E.g2_mul_batch(g2 * N, k[::-1].copy().reshape(-1), N)
p1 = np.asarray(p1).reshape(-1); p2 = np.asarray(p2).reshape(-1)

def best(f, reps=3):
    f(); b = 1e9
    for _ in range(reps):
        t = time.perf_counter(); f(); b = min(b, time.perf_counter() - t)
    return b * 1e3

msgs = [b"m%06d" % i for i in range(N)]
sks = k
pk, _ = E.g2_mul_generator_batch(sks.reshape(-1), N)
h = E.hash_g1_batch(msgs)
sg, _ = E.g1_mul_batch(np.asarray(h).reshape(-1), sks.reshape(-1), N)
pk = np.asarray(pk).reshape(-1); sg = np.asarray(sg).reshape(-1)
for n in (1024, 2048, 4096, 8192, 16384, 32768):
    row = []
    for thr in (1 << 20, 0):
        E.set_latency_threshold(thr)
        tp = best(lambda: E.pairing_batch(p1[:96 * n], p2[:192 * n], n))
        tv = best(lambda: E.g2pubs_verify_batch(msgs[:n], pk[:192 * n], sg[:96 * n]))
        row.append((tp, tv))
    print("n=%6d  pairing: latency path %7.2f ms  throughput kernels %7.2f ms   |  g2pubs verify: %7.2f ms  %7.2f ms" % (n, row[0][0], row[1][0], row[0][1], row[1][1]), flush=True)
