"""per-kernel breakdown of single-tuple calls from the library's own event log (blsmi_last_profile)"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from bls_amd import engine, _native
from oracle import refcpu as RC
engine.init(0); lib = _native.load()
sk = (12345).to_bytes(32, "big"); msg = b"Hello world! 16 characters 0"
pk, sig = RC.g2pubs.priv_to_pub(sk), RC.g2pubs.sign(msg, sk)
pk1, sig1 = RC.g1pubs.priv_to_pub(sk), RC.g1pubs.sign(msg, sk)
def run(name, fn):
    for _ in range(3): fn()
    b = 1e9
    for _ in range(10):
        t = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t)
    prof = bench.profiled(lib, fn)
    print("%-28s %.3f ms  " % (name, b * 1e3), {k: round(v[0], 3) for k, v in prof.items()}, "sum=%.3f" % sum(v[0] for v in prof.values()))
run("g2pubs verify n=1", lambda: engine.g2pubs_verify_batch([msg], pk, sig))
run("g1pubs verify n=1", lambda: engine.g1pubs_verify_batch([msg], pk1, sig1))
run("pairing n=1", lambda: engine.pairing_batch(sig, pk, 1))
run("hash_g1 n=1", lambda: engine.hash_g1_batch([msg]))
run("g1 mul n=1", lambda: engine.g1_mul_batch(sig, sk, 1))
run("g2 mul gen n=1", lambda: engine.g2_mul_generator_batch(sk, 1))
run("g2pubs verify n=64", lambda: engine.g2pubs_verify_batch([msg] * 64, pk * 64, sig * 64))
