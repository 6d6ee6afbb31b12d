"""configs[0] shape: ONE host call of 1 000 g2pubs tuples (and 1 000 g1pubs), best of 7, with the kernels' HIP-event times.  Environment decides the variant
(BLSMI_SIG_SIDE_MAX=1024: the signature side's Miller loop beside the hash also at this size).   python tools/cfg0_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from bls_amd import engine as E, _native
E.init(0)
lib = _native.load()
for group in ("g2pubs", "g1pubs"):
    n = 1000
    packed, pks, sigs = bench._verify_tuples(E, group, 1024, tag=7, nk=64)
    msgs = [bytes(packed.buf[int(packed.off[i]):int(packed.off[i + 1])]) for i in range(n)]
    pm = E.PackedMsgs(msgs)
    fn = E.g2pubs_verify_batch if group == "g2pubs" else E.g1pubs_verify_batch
    a, b = pks[:n].reshape(-1).copy(), sigs[:n].reshape(-1).copy()
    best = 1e9
    for _ in range(7):
        t0 = time.perf_counter(); ok, _ = fn(pm, a, b); best = min(best, time.perf_counter() - t0)
    assert all(ok)
    prof = bench.profiled(lib, lambda: fn(pm, a, b))
    print("%s 1000 tuples: %.3f ms/call (%.0f verifies/s) SIG_SIDE_MAX=%s %s" % (group, best * 1e3, n / best, os.environ.get("BLSMI_SIG_SIDE_MAX", "default"), {k: round(v[0], 3) for k, v in prof.items()}), flush=True)
