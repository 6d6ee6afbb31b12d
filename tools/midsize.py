"""Mid-size batches on the three pairing paths (one tuple per wave / lane quad / lane pair), resident inputs: wall time per call and
the per-kernel HIP-event times, n = 4 096 .. 65 536 (the hole between the latency hand-over and a full chip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bls_amd import engine as E, _native

E.init(0)
lib = _native.load()
dev = torch.device("cuda", 0)
N = 65536
g1, g2 = bench.synth_inputs(E, N, seed=5)
d1 = torch.from_numpy(g1).to(dev); d2 = torch.from_numpy(g2).to(dev); do = torch.zeros((N, 72), dtype=torch.int64, device=dev)
sizes = [int(x) for x in sys.argv[1:]] or [4096, 8192, 12288, 16384, 20480, 24576, 32768, 65536]
print("n, then ms per call (best of 3) and kernel ms: latency path | lane quad | lane pair")
for n in sizes:
    row = []
    for name, lat, quad in (("lat", 1 << 20, 0), ("quad", 0, 1 << 20), ("pair", 0, 0)):
        E.set_latency_threshold(lat); E.set_quad_threshold(quad)
        def step():
            E.pairing_batch_dev(d1.data_ptr(), d2.data_ptr(), do.data_ptr(), n)
        step(); step()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        prof = bench.profiled(lib, step)
        row.append("%s %.2f ms (%.2f M/s) %s" % (name, best * 1e3, n / best / 1e6, {k: round(v[0], 2) for k, v in prof.items() if not k.startswith("(")}))
    print("n=%6d  " % n + "  |  ".join(row), flush=True)
E.set_latency_threshold(8192); E.set_quad_threshold(16384)

# ---- verifies (g2pubs and g1pubs, hash-to-curve included) and one n-message g2pubs VerifyAggregate on the same three paths
for group in ("g2pubs", "g1pubs"):
    packed, pks, sigs = bench._verify_tuples(E, group, 32768, tag=3)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (packed.buf.copy(), packed.off.view(np.int64), pks, sigs)]
    d_ok = torch.zeros(32768, dtype=torch.uint8, device=dev)
    for n in [x for x in sizes if x <= 32768]:
        row = []
        for name, lat, quad in (("lat", 1 << 20, 0), ("quad", 0, 1 << 20), ("pair", 0, 0)):
            E.set_latency_threshold(lat); E.set_quad_threshold(quad)
            def step():
                E.verify_batch_dev(group, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, d_ok.data_ptr(), n)
            step(); step()
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            assert bool(d_ok[:n].all().item())
            prof = bench.profiled(lib, step)
            row.append("%s %.2f ms (%.2f M/s) %s" % (name, best * 1e3, n / best / 1e6, {k: round(v[0], 2) for k, v in prof.items() if not k.startswith("(")}))
        print("%s verify n=%6d  " % (group, n) + "  |  ".join(row), flush=True)
E.set_latency_threshold(8192); E.set_quad_threshold(16384)
for n in (16384, 32768):
    packed, allpk, agg, _ = bench._aggregate_inputs(E, "g2pubs", 0, n)
    d_m = torch.from_numpy(packed.buf.copy()).to(dev); d_o = torch.from_numpy(packed.off.view(np.int64).copy()).to(dev); d_k = torch.from_numpy(allpk).to(dev)
    for name, quad in (("quad", 16384), ("pair", 0)):
        E.set_quad_threshold(quad)
        def step():
            assert E.verify_aggregate_dev("g2pubs", d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), agg, n) is True
        step(); step()
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        prof = bench.profiled(lib, step)
        print("g2pubs VerifyAggregate n=%6d %s %.2f ms %s" % (n, name, best * 1e3, {k: round(v[0], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:4]}), flush=True)
E.set_quad_threshold(16384)
