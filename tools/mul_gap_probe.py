"""Where the 5 ms per step of the resident 2^20-point G1 scalar multiplication go that no kernel owns: wall time per
call against the kernel's own HIP-event duration, at several sizes.  Run under different HSA_SCRATCH_* settings to test the
hypothesis that the runtime allocates and releases the kernel's large scratch backing around every dispatch."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bls_amd import engine, _native

engine.init(0)
lib = _native.load()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(3)
N = 1 << 20
k = rng.integers(0, 256, size=(N, 32), dtype=np.uint8); k[:, 0] &= 0x3f
d_k = torch.from_numpy(k.reshape(-1)).to(dev)
for grp, pb in (("g1", 96), ("g2", 192)):
    bk = rng.integers(0, 256, size=(4096, 32), dtype=np.uint8); bk[:, 0] &= 0x3f
    bpts, _ = (engine.g1_mul_generator_batch if grp == "g1" else engine.g2_mul_generator_batch)(bk.reshape(-1), 4096)
    d_p = torch.from_numpy(np.ascontiguousarray(np.tile(bpts, (N // 4096, 1))).reshape(-1)).to(dev)
    d_out = torch.empty(N * pb, dtype=torch.uint8, device=dev); d_inf = torch.empty(N, dtype=torch.uint8, device=dev)
    for n in (1 << 20, 1 << 18, 1 << 16):
        def step():
            engine.mul_batch_dev(grp, d_p.data_ptr(), d_k.data_ptr(), d_out.data_ptr(), d_inf.data_ptr(), n)
        step(); step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
        prof = bench.profiled(lib, step)
        print("%s n=%7d  wall %.3f ms/call   kernels %s   env %s" % (grp, n, wall, {a: round(b[0], 3) for a, b in prof.items()},
              {e: os.environ[e] for e in os.environ if e.startswith("HSA_SCRATCH") or e.startswith("HSA_NO_SCRATCH")}), flush=True)
