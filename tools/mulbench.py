import sys, time, numpy as np
sys.path.insert(0, ".")
import bench
from bls_amd import engine
engine.init(0)
g1, g2 = bench._gens()
n = 1 << 20
rng = np.random.default_rng(1)
k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f
for name, fn, g in [("g1", engine.g1_mul_batch, g1), ("g2", engine.g2_mul_batch, g2)]:
    base, _ = fn(g * 4096, k[:4096].reshape(-1), 4096)
    pts = np.tile(base, (n // 4096, 1)).reshape(-1)
    fn(pts[:len(g) * 1024], k[:1024].reshape(-1), 1024)
    t = time.time(); out, inf = fn(pts, k.reshape(-1), n); dt = time.time() - t
    print(name, "1M scalar muls (host buffers incl. PCIe): %.3f s -> %.0f /s" % (dt, n / dt))
    for m in (1 << 20, 1 << 17, 1 << 14):
        msm = engine.g1_msm if name == "g1" else engine.g2_msm
        msm(pts[:len(g) * 4096], k[:4096].reshape(-1), 4096)
        best = 1e9
        for _ in range(3):
            t = time.time(); r = msm(pts[:len(g) * m], k[:m].reshape(-1), m); best = min(best, time.time() - t)
        print(name, "msm n=%d (host buffers incl. PCIe): %.1f ms -> %.0f points/s" % (m, best * 1e3, m / best))
