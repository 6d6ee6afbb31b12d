"""One bucket-method MSM of n points (G1 and G2) for profiling: rocprofv3 --kernel-trace --stats -- python tools/msm_probe.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from bls_amd import engine
engine.init(0)
g1, g2 = bench._gens()
rng = np.random.default_rng(1)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
k = rng.integers(0, 256, size=(m, 32), dtype=np.uint8); k[:, 0] &= 0x3f
for name, mul, msm, g, pb in (("g1", engine.g1_mul_batch, engine.g1_msm, g1, 96), ("g2", engine.g2_mul_batch, engine.g2_msm, g2, 192)):
    base, _ = mul(g * 4096, k[:4096].reshape(-1), 4096)
    pts = np.tile(base, (max(1, m // 4096), 1)).reshape(-1)[:pb * m]
    for _ in range(2):
        t = time.time(); msm(pts, k.reshape(-1), m); print(name, "msm", m, "%.1f ms" % ((time.time() - t) * 1e3))
