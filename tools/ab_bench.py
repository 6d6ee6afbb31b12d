#!/usr/bin/env python3
"""A/B timing of library variants: python tools/ab_bench.py libA.so libB.so ...  (each a build of blsmi.hip).
Runs the 64k-pairing device entry point 5 times per variant in separate subprocesses and prints ms/step."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import ctypes as C, sys, time, numpy as np, torch
lib = C.CDLL(sys.argv[1]); lib.blsmi_init(0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
sys.path.insert(0, %r)
import bench
from bls_amd import engine, _native
_native._lib = lib
g1, g2 = bench.synth_inputs(engine, n, 0)
dev = torch.device("cuda", 0)
a = torch.from_numpy(g1).to(dev); b = torch.from_numpy(g2).to(dev); o = torch.zeros((n, 72), dtype=torch.int64, device=dev)
lib.blsmi_set_profiling(1)
ts = []
for i in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    rc = lib.blsmi_pairing_batch_dev(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(o.data_ptr()), C.c_size_t(n), None)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    x, y = C.c_float(0), C.c_float(0); lib.blsmi_last_kernel_ms(C.byref(x), C.byref(y))
print("%%s rc=%%d best %%.3f ms/step (miller %%.3f, fexp %%.3f) -> %%.0f pairings/s checksum %%d" %% (sys.argv[1], rc, min(ts[1:]) * 1e3, x.value, y.value, n / min(ts[1:]), int(o[::997].sum().item()) & 0xffffffff))
''' % ROOT
sizes = [a for a in sys.argv[1:] if a.isdigit()] or ["65536"]
for so in [a for a in sys.argv[1:] if not a.isdigit()]:
    for n in sizes:
        subprocess.run([sys.executable, "-c", code, os.path.abspath(so), n], check=False)
