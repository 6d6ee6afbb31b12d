"""Throughput of the *WithDomain path (HashG2WithDomain + g1pubs verify), host buffers.  GPU box only."""
import sys, time, hashlib
sys.path.insert(0, ".")
import numpy as np
from bls_amd import engine
engine.init(0)
n = 65536
msgs = [hashlib.sha256(i.to_bytes(4, "little")).digest() for i in range(n)]
dom = bytes([1, 0, 0, 0, 0, 0, 0, 0])
engine.hash_g2_with_domain_batch(msgs[:256], dom)
t0 = time.perf_counter(); h = engine.hash_g2_with_domain_batch(msgs, dom); dt = time.perf_counter() - t0
print("hash_g2_with_domain n=%d: %.1f ms -> %.0f /s  checksum %d" % (n, dt * 1e3, n / dt, int(h.astype(np.uint64).sum())))
