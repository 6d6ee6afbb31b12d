"""Single-call latency of the small-batch path: Pairing / g2pubs.Verify / g1pubs.Verify with host buffers (GPU box only)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from bls_amd import engine
from oracle import refcpu as RC
engine.init(0)
sk = (12345).to_bytes(32, "big"); sk2 = (987654321).to_bytes(32, "big")
p1 = RC.g1_mul(RC.g1_generator(), sk); q1 = RC.g2_mul(RC.g2_generator(), sk2)
msg = b">16 character identical message"
def best(fn, reps=7):
    fn(); b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t0)
    return b * 1e3
for n in (1, 64, 1024, 4096):
    G1 = p1 * n; G2 = q1 * n
    print("pairing_batch n=%5d: %8.3f ms" % (n, best(lambda: engine.pairing_batch(G1, G2, n))))
pk = RC.g2pubs.priv_to_pub(sk); sig = RC.g2pubs.sign(msg, sk)
print("g2pubs verify n=1: %.3f ms  (hash_g1 alone: %.3f ms)" % (best(lambda: engine.g2pubs_verify_batch([msg], pk, sig)), best(lambda: engine.hash_g1_batch([msg]))))
pk1 = RC.g1pubs.priv_to_pub(sk); sig1 = RC.g1pubs.sign(msg, sk)
print("g1pubs verify n=1: %.3f ms  (hash_g2 alone: %.3f ms)" % (best(lambda: engine.g1pubs_verify_batch([msg], pk1, sig1)), best(lambda: engine.hash_g2_batch([msg]))))
for n in (64, 1024, 4096):
    ms = [msg] * n
    print("g2pubs verify n=%5d: %8.3f ms" % (n, best(lambda: engine.g2pubs_verify_batch(ms, pk * n, sig * n), 3)))
assert engine.g2pubs_verify_batch([msg], pk, sig)[0][0] and engine.g1pubs_verify_batch([msg], pk1, sig1)[0][0]
dom = bytes([42, 0, 0, 0, 0, 0, 0, 0]); m32 = b"Some msg".ljust(32, b"\0")
sgd = RC.g1pubs.sign_with_domain(m32, sk, dom)
assert engine.g1pubs_verify_with_domain_batch([m32], dom, pk1, sgd)[0]
print("g1pubs verify_with_domain n=1: %.3f ms  (hash_g2_with_domain alone: %.3f ms)" % (best(lambda: engine.g1pubs_verify_with_domain_batch([m32], dom, pk1, sgd)), best(lambda: engine.hash_g2_with_domain_batch([m32], dom))))
pkc = engine.g2_compress_batch(pk, 1); sgc = engine.g1_compress_batch(sig, 1)
for chk in (True, False):
    ok, ep, es = engine.verify_serialized_batch("g2pubs", [msg], pkc.reshape(-1), sgc.reshape(-1), chk)
    assert ok[0]
    print("g2pubs verify_serialized n=1 subgroup_check=%s: %.3f ms" % (chk, best(lambda: engine.verify_serialized_batch("g2pubs", [msg], pkc.reshape(-1), sgc.reshape(-1), chk))))
print("g2_decompress n=1 (checked): %.3f ms ; g1_decompress n=1 (checked): %.3f ms" % (best(lambda: engine.g2_decompress_batch(pkc.reshape(-1), 1, True)), best(lambda: engine.g1_decompress_batch(sgc.reshape(-1), 1, True))))
print("g1_sum n=128: %.3f ms ; g2_sum n=128: %.3f ms" % (best(lambda: engine.g1_sum(pk1 * 128, 128)), best(lambda: engine.g2_sum(pk * 128, 128))))
