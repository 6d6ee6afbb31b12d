"""Do two VerifyAggregate calls that run side by side on two streams (their hash and Miller-loop kernels interleaving on the SIMDs) finish
sooner than one after the other?  2 x 2^19 messages resident."""
import hashlib, sys, threading, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from bls_amd import engine as eng
eng.init(0)
dev = torch.device("cuda", 0)
nk, N = 256, 1 << 19
sk = b"".join(hashlib.sha256(b"ov-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
pks, _ = eng.g2_mul_generator_batch(sk, nk)
def case(tag):
    msgs = [hashlib.sha256(b"%s%d" % (tag, i)).digest() for i in range(N)]
    h = eng.hash_g1_batch(eng.PackedMsgs(msgs))
    sigs, _ = eng.g1_mul_batch(h.reshape(-1), sk * (N // nk), N)
    agg = eng.g1_sum(sigs.reshape(-1), N)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return (t(np.frombuffer(b"".join(msgs), dtype=np.uint8).copy()), t((np.arange(N + 1, dtype=np.uint64) * 32).view(np.int64)), t(np.tile(pks, (N // nk, 1)).reshape(-1)), agg)
A, B = case(b"a"), case(b"b")
def run(c, out, k):
    out[k] = eng.verify_aggregate_dev("g2pubs", c[0].data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3], N)
res = {}
run(A, res, 0); run(B, res, 1)
for rep in range(3):
    t0 = time.perf_counter(); run(A, res, 0); run(B, res, 1); seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(c, res, k)) for k, c in enumerate((A, B))]
    th[0].start(); time.sleep(0.012); th[1].start()                      # the second call starts when the first has finished hashing
    for x in th: x.join()
    par = time.perf_counter() - t0
    print("sequential %.1f ms, side by side %.1f ms  %s" % (seq * 1e3, par * 1e3, res))
