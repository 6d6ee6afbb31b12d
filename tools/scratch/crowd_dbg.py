import os, sys, threading, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_common import P, RC
from bls_amd import engine as E
E.init(0)
xs = P.XORShift(4096)
sks = [P.rand_fr(xs).to_bytes(32, "big") for _ in range(16)]
g1a = [RC.g2pubs.sign(b"x%d" % i, s) for i, s in enumerate(sks)]
g2a = [RC.g2pubs.priv_to_pub(s) for s in sks]
lib = E._lib()
def run(n, T, calls, log):
    a1 = np.frombuffer((b"".join(g1a) * (n // 16 + 1))[:96 * n], dtype=np.uint8); a2 = np.frombuffer((b"".join(g2a) * (n // 16 + 1))[:192 * n], dtype=np.uint8)
    bar = threading.Barrier(T + 1)
    def work(k):
        bar.wait()
        for c in range(calls):
            if log: lib.blsmi_set_profiling(1)
            E.pairing_batch(a1, a2, n)
            if log:
                buf = ctypes.create_string_buffer(4096); lib.blsmi_last_profile(buf, ctypes.c_size_t(4096))
                names = [x.split("=")[0] for x in buf.value.decode().split(";") if x]
                marks[k].append(names[0] if names else "?")
        bar.wait()
    marks = [[] for _ in range(T)]
    ts = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    for t in ts: t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t in ts: t.join()
    lib.blsmi_set_profiling(0)
    return dt / calls * 1e3, marks
for n in (2048,):
    for label, crowd, assume in (("own size", 0, 0), ("crowd", 1, 0), ("assumed load", 1, 100000)):
        E.set_option("crowd_quad", crowd); E.set_option("assume_load", assume)
        run(n, 4, 3, False)
        ms, _ = run(n, 4, 12, False)
        _, marks = run(n, 4, 12, True)
        print(n, label, "%.3f ms per call" % ms, [",".join(m[:12]) for m in marks][:2])
E.set_option("assume_load", 0)
