"""Mid-size VerifyAggregate (distinct messages), resident inputs: ms per call with the lane-row Miller kernel against the one-tuple-per-wave
programs (round 6).   python tools/agg_mid.py [sizes ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bls_amd import engine as E, _native
E.init(0)
lib = _native.load()
dev = torch.device("cuda", 0)
sizes = [int(x) for x in sys.argv[1:]] or [2304, 4096, 8192]
for group in ("g2pubs", "g1pubs"):
    for n in sizes:
        packed, allpk, agg, _ = bench._aggregate_inputs(E, group, 0, n)
        d_m = torch.from_numpy(packed.buf.copy()).to(dev); d_o = torch.from_numpy(packed.off.view(np.int64).copy()).to(dev); d_k = torch.from_numpy(allpk).to(dev)
        row = []
        for name, rw in (("wave", (0, 0)), ("row", E.ROW_DEFAULT)):
            E.set_row_threshold(*rw)
            def step():
                assert E.verify_aggregate_dev(group, d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), agg, n) is True
            step(); step()
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            prof = bench.profiled(lib, step)
            row.append("%s %.2f ms %s" % (name, best * 1e3, {k.replace("k_", ""): round(v[0], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:5] if not k.startswith("(")}))
        print("%s VerifyAggregate n=%5d  " % (group, n) + "  |  ".join(row), flush=True)
E.set_row_threshold(*E.ROW_DEFAULT)
