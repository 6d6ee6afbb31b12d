"""A/B of the level program `pairing1s` (straight-line) against `pairing1` (its squaring runs as K_REP loops), same box, interleaved:
lone bls.Pairing latency through the host entry point, the kernel's own duration (HIP events), and the batch sizes up to the latency
path's hand-over.  VERDICT r04 item 8: measure the repeat construct, then adopt or close.   python tools/rolled_ab.py"""
import ctypes
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bls_amd import engine, _native  # noqa: E402
from oracle import refcpu as RC  # noqa: E402

engine.init(0)
lib = _native.load()
g1, g2 = bench.synth_inputs(engine, 4096, seed=7)
want = RC.pairing_batch(g1[:4].tobytes(), g2[:4].tobytes(), 4)
for n in (1, 64, 1024, 4096):
    a, b = g1[:n].reshape(-1), g2[:n].reshape(-1)
    res = {0: [], 1: []}
    kern = {0: [], 1: []}
    for rnd in range(30 if n == 1 else 8):
        for mode in (0, 1):
            engine.set_option("lat_rolled", mode)
            t0 = time.perf_counter(); out = engine.pairing_batch(a, b, n); res[mode].append(time.perf_counter() - t0)
            assert np.array_equal(out[:min(n, 4)], want[:min(n, 4)]), ("parity", n, mode)
            if rnd % 4 == 0:
                prof = bench.profiled(lib, lambda: engine.pairing_batch(a, b, n))
                kern[mode].append(sum(v[0] for k, v in prof.items() if k.startswith("k_lat:pairing1")))
    for mode in (0, 1):
        r = sorted(res[mode])
        print("n=%5d  %-10s  call min %.3f ms  median %.3f ms   kernel median %.3f ms" % (n, "rolled" if mode else "straight", r[0] * 1e3, statistics.median(r) * 1e3, statistics.median(kern[mode])))
engine.set_option("lat_rolled", 0)

hdr = open(os.path.join(ROOT, "bls_amd", "csrc", "lat_programs.h")).read()
import re  # noqa: E402
sz = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define LAT_(\w+)_BYTES (\d+)", hdr)}
print("program bytes: straight %d, rolled %d (%.0f %%)" % (sz["PAIRING1S"], sz["PAIRING1"], 100.0 * sz["PAIRING1"] / sz["PAIRING1S"]))
