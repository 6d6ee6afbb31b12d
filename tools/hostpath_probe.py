"""host-buffer entry points with logical shards on one device (copies of one block beside another block's kernels)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from bls_amd import engine
engine.init_devices(1)
class E: pass
E.engine = engine
r = bench.inlibrary_bench(E, 1)
print("SHARDS=%s" % os.environ.get("BLSMI_SHARDS", "1"), {k: v for k, v in r.items() if k.endswith("per_s") or k.endswith("ms_per_call")})
n = 1 << 20
rng = np.random.default_rng(1)
k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f
g1, g2 = bench._gens()
base, _ = engine.g1_mul_generator_batch(k[:4096].reshape(-1), 4096)
pts = np.ascontiguousarray(np.tile(base, (n // 4096, 1))).reshape(-1)
for name, fn in (("g1_mul 1M host", lambda: engine.g1_mul_batch(pts, k.reshape(-1), n)), ("g1_msm 1M host", lambda: engine.g1_msm(pts, k.reshape(-1), n))):
    fn(); b = 1e9
    for _ in range(3):
        t = time.perf_counter(); fn(); b = min(b, time.perf_counter() - t)
    print(name, "%.2f ms" % (b * 1e3))
