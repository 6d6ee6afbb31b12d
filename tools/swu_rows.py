"""The SWU maps of a hash a lane per map (k_swu_g?_two_lanes) against a row of sixteen lanes per map (k_swu_g?_rows), inside a lone caller's Verify of n tuples,
resident inputs, every other threshold as shipped: wall time per call (best of 5) and the per-kernel HIP-event times.
    python tools/swu_rows.py [sizes ...]        (default 600 1000 1536 2048 3072 4096 6144 8192)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bls_amd import engine as E, _native

E.init(0)
lib = _native.load()
dev = torch.device("cuda", 0)
sizes = [int(x) for x in sys.argv[1:]] or [600, 1000, 1536, 2048, 3072, 4096, 6144, 8192]
nmax = max(sizes)


def timed(step, reps=5):
    step(); step()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


for pkg in ("g2pubs", "g1pubs"):
    packed, pks, sigs = bench._verify_tuples(E, pkg, nmax, tag=3)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (packed.buf.copy(), packed.off.view(np.int64), pks, sigs)]
    d_ok = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    for n in sizes:
        row = []
        for name, mx in (("lanes", 0), ("rows", 1 << 20)):
            E.set_option("swu_row_max", mx)
            def step():
                E.verify_batch_dev(pkg, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, d_ok.data_ptr(), n)
            best = timed(step)
            assert bool(d_ok[:n].all().item())
            prof = bench.profiled(lib, step)
            row.append("%s %.2f ms (%.2f M/s) %s" % (name, best * 1e3, n / best / 1e6, {k.replace("k_", ""): round(v[0], 2) for k, v in prof.items() if "hash" in k or "swu" in k or "clear" in k}))
        print("%s verify n=%6d  " % (pkg, n) + "  |  ".join(row), flush=True)
