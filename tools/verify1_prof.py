"""One g2pubs.Verify per call, repeated: for rocprofv3 --kernel-trace (kernel time) against the wall time per call."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bls_amd import engine
from oracle import refcpu as RC
engine.init(0)
sk = (12345).to_bytes(32, "big")
msg = b"Hello world! 16 characters 0"
pk, sig = RC.g2pubs.priv_to_pub(sk), RC.g2pubs.sign(msg, sk)
for _ in range(5):
    engine.g2pubs_verify_batch([msg], pk, sig)
t = time.perf_counter()
R = 50
for _ in range(R):
    ok = engine.g2pubs_verify_batch([msg], pk, sig)[0][0]
print("verify n=1: %.3f ms per call (wall), ok=%s" % ((time.perf_counter() - t) / R * 1e3, ok))
