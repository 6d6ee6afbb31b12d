"""A lone caller's mid-size batches on the FOUR pairing layouts (one tuple per wave / per lane row / per lane quad / per lane pair), resident
inputs: wall time per call (best of 5) and the per-kernel HIP-event times.  VERDICT r05 item 1: 1 024 .. 8 192 tuples.
    python tools/midsize4.py [sizes ...]        (default 1024 2048 3072 4096 6144 8192 12288 16384)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bls_amd import engine as E, _native

E.init(0)
lib = _native.load()
dev = torch.device("cuda", 0)
N = 16384
sizes = [int(x) for x in sys.argv[1:]] or [1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384]
N = max(N, max(sizes))
g1, g2 = bench.synth_inputs(E, N, seed=5)
d1 = torch.from_numpy(g1).to(dev); d2 = torch.from_numpy(g2).to(dev); do = torch.zeros((N, 72), dtype=torch.int64, device=dev)
LAYOUTS = (("wave", 1 << 20, 0, (0, 0)), ("row", 8192, 16384, (1, 1 << 20)), ("quad", 0, 1 << 20, (0, 0)), ("pair", 0, 0, (0, 0)))
if os.environ.get("MIDSIZE_ROW_ONLY"):
    LAYOUTS = LAYOUTS[:2]


def timed(step, reps=5):
    step(); step()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


print("pairings: n, then ms per call (best of 5), M/s and kernel ms per layout")
for n in sizes:
    row = []
    for name, lat, quad, rw in LAYOUTS:
        if name == "wave" and n > 8192:
            continue
        E.set_latency_threshold(lat); E.set_quad_threshold(quad); E.set_row_threshold(*rw)
        def step():
            E.pairing_batch_dev(d1.data_ptr(), d2.data_ptr(), do.data_ptr(), n)
        best = timed(step)
        prof = bench.profiled(lib, step)
        row.append("%s %.2f ms (%.2f M/s) %s" % (name, best * 1e3, n / best / 1e6, {k.replace("k_", ""): round(v[0], 2) for k, v in prof.items() if not k.startswith("(")}))
    print("n=%6d  " % n + "  |  ".join(row), flush=True)

for group in ("g2pubs", "g1pubs"):
    nmax = max(sizes)
    packed, pks, sigs = bench._verify_tuples(E, group, nmax, tag=3)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (packed.buf.copy(), packed.off.view(np.int64), pks, sigs)]
    d_ok = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    for n in sizes:
        row = []
        for name, lat, quad, rw in LAYOUTS + (("row-noside", 8192, 16384, (1, 1 << 20)),):
            if name == "wave" and n > 8192:
                continue
            E.set_latency_threshold(lat); E.set_quad_threshold(quad); E.set_row_threshold(*rw); E.set_option("row_side", 0 if name == "row-noside" else 1)
            def step():
                E.verify_batch_dev(group, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, d_ok.data_ptr(), n)
            best = timed(step)
            assert bool(d_ok[:n].all().item())
            prof = bench.profiled(lib, step)
            row.append("%s %.2f ms (%.2f M/s) %s" % (name, best * 1e3, n / best / 1e6, {k.replace("k_", ""): round(v[0], 2) for k, v in prof.items() if not k.startswith("(")}))
        print("%s verify n=%6d  " % (group, n) + "  |  ".join(row), flush=True)
E.set_latency_threshold(8192); E.set_quad_threshold(16384); E.set_row_threshold(*E.ROW_DEFAULT)
