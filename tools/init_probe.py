"""Cost of bringing the library up in a fresh process: dlopen + blsmi_init (streams, generator tables, level programs), first calls."""
import sys, time
sys.path.insert(0, "/root/repo")
t0 = time.perf_counter()
from bls_amd import _native
lib = _native.load()
t1 = time.perf_counter()
from bls_amd import engine as eng
eng.init(0)
t2 = time.perf_counter()
from oracle import refcpu as RC
g1, g2 = RC.g1_generator(), RC.g2_generator()
t3 = time.perf_counter()
eng.pairing_batch(g1, g2, 1)
t4 = time.perf_counter()
eng.pairing_batch(g1, g2, 1)
t5 = time.perf_counter()
print("INIT load %.0f ms, blsmi_init %.0f ms, first pairing %.1f ms, second %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
