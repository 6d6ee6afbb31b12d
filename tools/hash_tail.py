"""HashG2's cofactor clearing on three layouts -- a lane pair per message (k_hash_g2_pair, the whole hash in one kernel), sixteen lanes (k_hash_g2_front +
k_clear_h2_row) or four lanes per message (k_hash_g2_front + k_clear_h2_quad) -- inside a lone caller's g1pubs Verify of n tuples, resident inputs, every
other threshold as shipped: wall time per call (best of 5) and the per-kernel HIP-event times.
    python tools/hash_tail.py [sizes ...]        (default 2048 4096 6144 8192 12288 16384 24576 32768)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bls_amd import engine as E, _native

E.init(0)
lib = _native.load()
dev = torch.device("cuda", 0)
sizes = [int(x) for x in sys.argv[1:]] or [2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768]
nmax = max(sizes)
TAILS = (("pair", (1, 0), (1, 0), (1, 0)), ("row", (1, 1 << 20), (1, 0), (1, 0)), ("quad", (1, 0), (1, 1 << 20), (1, 0)), ("oct", (1, 0), (1, 0), (1, 1 << 20)), ("shipped", (2048, 4096), (4097, 16384), (2048, 7168)))


def timed(step, reps=5):
    step(); step()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best


if os.environ.get("HASH_TAIL_G1"):                                         # HashG1's tail inside a g2pubs Verify: a message per lane / four lanes per message / the level program
    packed, pks, sigs = bench._verify_tuples(E, "g2pubs", nmax, tag=3)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (packed.buf.copy(), packed.off.view(np.int64), pks, sigs)]
    d_ok = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    for n in sizes:
        row = []
        for name, lo, hi in (("lane", 1, 0), ("quad", 1, 1 << 20), ("shipped", 1280, 32768)):
            E.set_option("hash_g1_quad_min", lo); E.set_option("hash_g1_quad_max", hi)
            def step():
                E.verify_batch_dev("g2pubs", d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, d_ok.data_ptr(), n)
            best = timed(step)
            assert bool(d_ok[:n].all().item())
            prof = bench.profiled(lib, step)
            row.append("%s %.2f ms (%.2f M/s) %s" % (name, best * 1e3, n / best / 1e6, {k.replace("k_", ""): round(v[0], 2) for k, v in prof.items() if "hash" in k or "swu" in k}))
        print("g2pubs verify n=%6d  " % n + "  |  ".join(row), flush=True)
    sys.exit(0)
packed, pks, sigs = bench._verify_tuples(E, "g1pubs", nmax, tag=3)
d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (packed.buf.copy(), packed.off.view(np.int64), pks, sigs)]
d_ok = torch.zeros(nmax, dtype=torch.uint8, device=dev)
for n in sizes:
    row = []
    for name, rw, qd, oc in TAILS:
        E.set_option("hash_row_min", rw[0]); E.set_option("hash_row_max", rw[1]); E.set_option("hash_quad_min", qd[0]); E.set_option("hash_quad_max", qd[1]); E.set_option("hash_oct_min", oc[0]); E.set_option("hash_oct_max", oc[1])
        def step():
            E.verify_batch_dev("g1pubs", d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, d_ok.data_ptr(), n)
        best = timed(step)
        assert bool(d_ok[:n].all().item())
        prof = bench.profiled(lib, step)
        row.append("%s %.2f ms (%.2f M/s) %s" % (name, best * 1e3, n / best / 1e6, {k.replace("k_", ""): round(v[0], 2) for k, v in prof.items() if "hash" in k or "clear" in k}))
    print("g1pubs verify n=%6d  " % n + "  |  ".join(row), flush=True)
