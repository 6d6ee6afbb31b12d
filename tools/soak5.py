"""Differential soak of the round-3 hash kernels at scale: HashG2 with a lane pair per message against the one-lane kernel (run the
script twice, with and without BLSMI_HASH_G2_PAIR=0: the digests must match), and HashG2WithDomain with the wave-shared search
against the latency path (eight lanes per message + level program) on the same messages, in process.
python tools/soak5.py [log2 n]   -> prints DIGEST lines and "domain ok"."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bls_amd import engine as eng  # noqa: E402

eng.init(0)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << lg
rng = np.random.default_rng(31337)
lens = rng.integers(0, 120, size=n)
blob = rng.integers(0, 256, size=int(lens.sum()), dtype=np.uint8).tobytes()
offs = np.concatenate([[0], np.cumsum(lens)])
msgs = [blob[offs[i]:offs[i + 1]] for i in range(n)]
eng.set_latency_threshold(0)
h = eng.hash_g2_batch(msgs)
print("DIGEST hash_g2 n=%d %s" % (n, hashlib.sha256(h.tobytes()).hexdigest()))
# HashG2WithDomain: throughput kernel (shared search) vs the latency path in chunks of 4096
m32 = [rng.bytes(32) for _ in range(1 << 16)]
dom = rng.bytes(8)
big = eng.hash_g2_with_domain_batch(m32, dom)
eng.set_latency_threshold(8192)
bad = 0
for lo in range(0, len(m32), 4096):
    small = eng.hash_g2_with_domain_batch(m32[lo:lo + 4096], dom)
    bad += int((small != big[lo:lo + 4096]).any(axis=1).sum())
print("DIGEST domain n=%d %s mismatches=%d" % (len(m32), hashlib.sha256(big.tobytes()).hexdigest(), bad))
assert bad == 0
print("domain ok")
