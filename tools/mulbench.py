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
