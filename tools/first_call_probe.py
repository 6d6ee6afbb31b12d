"""First and later latency of a 1 000-tuple g2pubs verify call in a fresh process (scratch growth, lazy kernel loading)."""
import sys, time, hashlib
sys.path.insert(0, "/root/repo")
import numpy as np
from bls_amd import engine as eng
eng.init(0)
n = 1000
sk = b"".join(hashlib.sha256(b"fc-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(64))
pks, _ = eng.g2_mul_generator_batch(sk, 64)
msgs = [b"Hello world! 16 characters %d" % i for i in range(n)]
h = eng.hash_g1_batch(msgs[:8])                       # small call first (waves kernels)
big = [b"m%d" % i for i in range(65536)]
t0 = time.perf_counter(); hb = eng.hash_g1_batch(eng.PackedMsgs(big)); print("hash 65536 first: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
allh = eng.hash_g1_batch(msgs)
sigs, _ = eng.g1_mul_batch(allh.reshape(-1), (sk * 16)[:32 * n], n)
allpk = np.tile(pks, (16, 1))[:n]
for rep in range(4):
    t0 = time.perf_counter(); ok, _ = eng.g2pubs_verify_batch(msgs, allpk.reshape(-1), sigs.reshape(-1)); dt = time.perf_counter() - t0
    print("verify 1000 call %d: %.2f ms  all ok %s" % (rep, dt * 1e3, bool(ok.all())))
