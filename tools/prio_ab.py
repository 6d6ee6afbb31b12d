"""A lone caller's Verify of n tuples, resident inputs: wall time per call (min and median of 9) and the per-kernel HIP-event times, g1pubs and g2pubs, for g2pubs with
the signature side on the side stream as well ("row_side_g2pubs").  Run once with the shipped library and once with BLSMI_LIB=ab/libblsmi_noprio.so
(tools/build_variant.sh noprio k_hash.hip,k_hash_pair.hip,k_hash_quad.hip,k_pairing_row.hip "-DBLSMI_HASH_PRIO=0"): the hash kernels with and without their raised wave priority.
    python tools/prio_ab.py [sizes ...]        (default 2048 3072 4096 6144 8192)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bls_amd import engine as E, _native

E.init(0)
lib = _native.load()
dev = torch.device("cuda", 0)
sizes = [int(x) for x in sys.argv[1:]] or [2048, 3072, 4096, 6144, 8192]
nmax = max(sizes)
print("library:", os.environ.get("BLSMI_LIB", "shipped"))


def timed(step, reps=9):
    step(); step()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


G1 = (("shipped", {}), ("oct tail", {"hash_oct_min": 1, "hash_oct_max": 1 << 20}), ("no oct tail", {"hash_oct_max": 0}), ("row tail", {"hash_oct_max": 0, "hash_row_min": 1, "hash_row_max": 1 << 20}), ("quad tail", {"hash_oct_max": 0, "hash_row_max": 0, "hash_quad_min": 1, "hash_quad_max": 1 << 20}), ("side kernel with 40 KB of LDS a workgroup", {"row_side_lds": 40960}))
G2 = (("shipped", {}), ("two-pair loop, maps in rows", {"row_side_g2pubs": 0}), ("side kernel with 40 KB of LDS a workgroup", {"row_side_lds": 40960}))
DEFAULTS = {"hash_row_min": 2048, "hash_row_max": 4096, "hash_quad_min": 4097, "hash_quad_max": 16384, "row_side_g2pubs": 1, "swu_row_max": 4096, "row_side_piece": 0, "row_side_lds": 0, "hash_oct_min": 2048, "hash_oct_max": 7168}
for pkg, variants in (("g1pubs", G1), ("g2pubs", G2)):
    packed, pks, sigs = bench._verify_tuples(E, pkg, nmax, tag=3)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (packed.buf.copy(), packed.off.view(np.int64), pks, sigs)]
    d_ok = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    for n in sizes:
        row = []
        for name, opts in variants:
            for k, v in DEFAULTS.items():
                E.set_option(k, v)
            for k, v in opts.items():
                E.set_option(k, v)
            def step():
                E.verify_batch_dev(pkg, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, d_ok.data_ptr(), n)
            best, med = timed(step)
            assert bool(d_ok[:n].all().item())
            prof = bench.profiled(lib, step)
            row.append("%s min %.2f / median %.2f ms (%.2f M/s) %s" % (name, best * 1e3, med * 1e3, n / best / 1e6, {k.replace("k_", ""): round(v[0], 2) for k, v in prof.items()}))
        print("%s verify n=%6d  " % (pkg, n) + "\n        ".join(row), flush=True)
for k, v in DEFAULTS.items():
    E.set_option(k, v)
