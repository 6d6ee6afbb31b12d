"""What callers that arrive TOGETHER get at the sizes that do not fill the chip alone (the mid-size regime).
T OS threads (what cgo gives goroutines), each issuing g2pubs verify calls of n tuples through the host entry point (in-memory points,
blsmi_g2pubs_verify_batch_jac) or pairing calls (blsmi_pairing_batch): the library leases each call its own stream (BLSMI_STREAMS, 4 per device),
so their kernels share the SIMDs.  Reports the aggregate rate against the lone caller's.  python tools/midsize_concurrency.py [sizes,comma,separated]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_common import P, RC, g1_to_jac, g2_to_jac  # noqa: E402
from bls_amd import engine as E  # noqa: E402

E.init(0)
xs = P.XORShift(4096)
NK = 64
sks = [P.rand_fr(xs).to_bytes(32, "big") for _ in range(NK)]
msgs = [b"mid-size %d" % i for i in range(NK)]
pkj = [g2_to_jac(RC.g2pubs.priv_to_pub(s), (3 + i, 5)) for i, s in enumerate(sks)]
sgj = [g1_to_jac(RC.g2pubs.sign(m, s), 7 + i) for i, (m, s) in enumerate(zip(msgs, sks))]
g1a = [RC.g2pubs.sign(m, s) for m, s in zip(msgs, sks)]
g2a = [RC.g2pubs.priv_to_pub(s) for s in sks]


def batch(n):
    r = (n + NK - 1) // NK
    return (E.PackedMsgs((msgs * r)[:n]), np.frombuffer((b"".join(pkj) * r)[:288 * n], dtype=np.uint8), np.frombuffer((b"".join(sgj) * r)[:144 * n], dtype=np.uint8),
            np.frombuffer((b"".join(g1a) * r)[:96 * n], dtype=np.uint8), np.frombuffer((b"".join(g2a) * r)[:192 * n], dtype=np.uint8))


def run(fn, threads, calls):
    bar = threading.Barrier(threads + 1)
    def work():
        bar.wait()
        for _ in range(calls):
            fn()
        bar.wait()
    ts = [threading.Thread(target=work) for _ in range(threads)]
    for t in ts:
        t.start()
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    return dt


SIZES = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256, 1024, 2048, 4096, 8192, 16384]
if os.environ.get("CROWD_FLOOR"):
    E.set_option("crowd_floor", int(os.environ["CROWD_FLOOR"]))        # (A/B of the floor itself)
print("%-8s %-6s %-8s %8s %14s %14s %10s" % ("shape", "mode", "tuples", "threads", "ms per call", "aggregate /s", "vs lone"))
for shape, crowd in (("verify", 0), ("verify", 1), ("verify", 2), ("pairing", 0), ("pairing", 1)):
    # 0: layout by the call's own size, mid-size calls never merged (the behaviour before this change); 1: layout by what the device carries;
    # 2 (the default): + concurrent mid-size Verify calls merge into one launch (verify_host.inc: the combiner's mid-size class)
    E.set_option("crowd_quad", 1 if crowd else 0); E.set_option("combine_mid_max", 8192 if crowd == 2 else 0)
    for n in SIZES:
        pm, pk, sg, a1, a2 = batch(n)
        if shape == "verify":
            fn = lambda: E.g2pubs_verify_batch_jac(pm, pk, sg)
            assert bool(np.all(fn()[0]))
        else:
            fn = lambda: E.pairing_batch(a1, a2, n)
        fn(); fn()
        lone = None
        for T in (1, 2, 4, 8):
            calls = max(3, min(40, 60000 // n))
            dt = min(run(fn, T, calls) for _ in range(2))
            rate = T * calls * n / dt
            lone = lone or rate
            print("%-8s %-6d %-8d %8d %14.3f %14.0f %10.2f" % (shape, crowd, n, T, dt / calls * 1e3, rate, rate / lone), flush=True)
