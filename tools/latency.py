"""Latency/throughput of blsmi_g2pubs_verify_batch (host buffers, PCIe included) against the batch size:
what one cgo Verify() call costs, and from which batch size the GPU is saturated.  GPU box only."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import bench
from bls_amd import engine

engine.init(0)
nmax = 65536
packed, pks, sigs = bench._verify_tuples(engine, "g2pubs", nmax)
buf, off = packed.buf, packed.off
for n in (1, 2, 32, 64, 1024, 8192, 65536):
    msgs = [bytes(buf[int(off[i]):int(off[i + 1])]) for i in range(n)]
    pk = pks[:n].tobytes(); sg = sigs[:n].tobytes()
    engine.g2pubs_verify_batch(msgs, pk, sg)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); ok, _ = engine.g2pubs_verify_batch(msgs, pk, sg); best = min(best, time.perf_counter() - t0)
    assert ok.all()
    print("n=%6d  %9.3f ms per call  %10.0f verifies/s" % (n, best * 1e3, n / best))

# concurrent single-tuple callers (one OS thread each, as cgo would): merged into shared launches by the library
import threading
for nthreads in (1, 16, 64, 256):
    reps = 6
    def work(i):
        m = [bytes(buf[int(off[i]):int(off[i + 1])])]
        for _ in range(reps):
            ok, _ = engine.g2pubs_verify_batch(m, pks[i].tobytes(), sigs[i].tobytes())
            assert ok[0]
    ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print("threads=%4d single-tuple calls: %8.0f verifies/s (%.1f ms per call)" % (nthreads, nthreads * reps / dt, dt / reps * 1e3))

# dump the tuples for tools/native/combine_bench (threads without an interpreter lock)
import struct, os
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/tuples.bin", "wb") as f:
    m = 1024
    f.write(struct.pack("<Q", m))
    for i in range(m):
        msg = bytes(buf[int(off[i]):int(off[i + 1])])
        f.write(struct.pack("<I", len(msg))); f.write(msg); f.write(pks[i].tobytes()); f.write(sigs[i].tobytes())
print("wrote gpurun_out/tuples.bin")

# Deserialize + Verify over the compressed wire format
from oracle import refcpu as RC  # test tooling only
n = 65536
pkc = engine.g2_compress_batch(pks[:n].reshape(-1), n); sgc = engine.g1_compress_batch(sigs[:n].reshape(-1), n)
msgs = [bytes(buf[int(off[i]):int(off[i + 1])]) for i in range(n)]
for chk in (True, False):
    engine.verify_serialized_batch("g2pubs", msgs[:1024], pkc[:1024].reshape(-1), sgc[:1024].reshape(-1), chk)
    t0 = time.perf_counter(); ok, ep, es = engine.verify_serialized_batch("g2pubs", msgs, pkc.reshape(-1), sgc.reshape(-1), chk); dt = time.perf_counter() - t0
    assert ok.all()
    print("verify_serialized g2pubs n=%d subgroup_check=%s: %.1f ms -> %.0f /s (host buffers)" % (n, chk, dt * 1e3, n / dt))
