import sys, json, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from bls_amd import engine, _native
engine.init(0); lib = _native.load()
n = 1 << 20
rng = np.random.default_rng(3)
k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f
dev = torch.device("cuda", 0)
d_k = torch.from_numpy(k.reshape(-1)).to(dev)
for grp, pb in (("g1", 96), ("g2", 192)):
    bk = rng.integers(0, 256, size=(4096, 32), dtype=np.uint8); bk[:, 0] &= 0x3f
    bpts, _ = (engine.g1_mul_generator_batch if grp == "g1" else engine.g2_mul_generator_batch)(bk.reshape(-1), 4096)
    d_p = torch.from_numpy(np.ascontiguousarray(np.tile(bpts, (n // 4096, 1))).reshape(-1)).to(dev)
    d_one = torch.zeros(pb, dtype=torch.uint8, device=dev)
    f = lambda: engine.msm_dev(grp, d_p.data_ptr(), d_k.data_ptr(), n, d_one.data_ptr())
    f(); f()
    import time
    t0 = time.perf_counter(); f(); f(); f(); dt = (time.perf_counter() - t0) / 3
    prof = bench.profiled(lib, f)
    print(grp, "msm ms %.2f" % (dt * 1e3), {a: round(b[0], 3) for a, b in sorted(prof.items(), key=lambda kv: -kv[1][0])})
    # the raw sequence of segments of one call
    bench.read_profile(lib); lib.blsmi_set_profiling(1); f(); lib.blsmi_set_profiling(0)
    buf = ctypes.create_string_buffer(1 << 16); lib.blsmi_last_profile(buf, ctypes.c_size_t(len(buf)))
    print(grp, "msm seq", buf.value.decode())
