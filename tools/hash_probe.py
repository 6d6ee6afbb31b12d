"""Time hash-to-curve batches on the GPU (tools/hash_probe.py [n]); meant to run under rocprofv3 --kernel-trace --stats."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bls_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
E.init()
rng = np.random.default_rng(5)
msgs = [rng.bytes(32) for _ in range(n)]
for name in ("hash_g1_batch", "hash_g2_batch"):
    f = getattr(E, name)
    f(msgs[:1024])
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); f(msgs); best = min(best, time.perf_counter() - t)
    print(f"{name}: n={n} {best*1e3:.2f} ms (host buffers, wall) {n/best/1e6:.3f} M/s")
dom = bytes(range(8))
E.hash_g2_with_domain_batch(msgs[:1024], dom)
best = 1e9
for _ in range(3):
    t = time.perf_counter(); E.hash_g2_with_domain_batch(msgs, dom); best = min(best, time.perf_counter() - t)
print(f"hash_g2_with_domain_batch: n={n} {best*1e3:.2f} ms (host buffers, wall) {n/best/1e6:.3f} M/s")
