"""config 4: one g2pubs.VerifyAggregate over n distinct 32-byte messages (host buffers), timed at the C ABI.  GPU box only."""
import os, sys, time, hashlib, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bls_amd import engine, _native
engine.init(0)
lib = _native.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
nk = 256
g1, g2 = bench._gens()
sk = b"".join(hashlib.sha256(b"agg-sk-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
idx = np.arange(n, dtype=np.uint64)
msgs = np.frombuffer(b"".join(hashlib.sha256(int(i).to_bytes(8, "little")).digest() for i in range(n)), dtype=np.uint8).copy()
off = (np.arange(n + 1, dtype=np.uint64) * 32)
pks, _ = engine.g2_mul_batch(g2 * nk, sk, nk)
allpk = np.ascontiguousarray(np.tile(pks, (n // nk, 1)))
u8p = C.POINTER(C.c_uint8); u64p = C.POINTER(C.c_uint64)
h = np.zeros((n, 96), dtype=np.uint8)
assert lib.blsmi_hash_g1_batch(msgs.ctypes.data_as(u8p), off.ctypes.data_as(u64p), h.ctypes.data_as(u8p), C.c_size_t(n)) == 0
sigs, _ = engine.g1_mul_batch(h.reshape(-1), sk * (n // nk), n)
agg = engine.g1_sum(sigs.reshape(-1), n)
sig = np.frombuffer(agg, dtype=np.uint8).copy()
ok = C.c_int(0)
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.blsmi_g2pubs_verify_aggregate(msgs.ctypes.data_as(u8p), off.ctypes.data_as(u64p), allpk.ctypes.data_as(u8p), sig.ctypes.data_as(u8p), C.c_size_t(n), C.byref(ok))
    dt = time.perf_counter() - t0
    print("VerifyAggregate n=%d: rc=%d ok=%d  %.1f ms -> %.0f signatures/s" % (n, rc, ok.value, dt * 1e3, n / dt))
