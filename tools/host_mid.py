"""A lone caller's mid-size Verify from HOST buffers (what a cgo caller does) against the same call on resident inputs: ms per call (best of 7), affine and in-memory
(Jacobian) point forms, with the library's per-kernel HIP-event times of the host call.    python tools/host_mid.py [sizes ...]   (default 1000 2048 4096 8192)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import bench
from bls_amd import engine as E, _native
from gpu_common import g1_to_jac, g2_to_jac

E.init(0)
lib = _native.load()
dev = torch.device("cuda", 0)
sizes = [int(x) for x in sys.argv[1:]] or [1000, 2048, 4096, 8192]
nmax = max(sizes)


def best(fn, reps=7):
    fn(); fn()
    b = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b * 1e3


for pkg in ("g2pubs", "g1pubs"):
    packed, pks, sigs = bench._verify_tuples(E, pkg, nmax, tag=5, nk=64)
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (packed.buf.copy(), packed.off.view(np.int64), pks, sigs)]
    d_ok = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    g2 = pkg == "g2pubs"
    base_pk = [(g2_to_jac if g2 else g1_to_jac)(pks[i].tobytes(), ((3 + i, 2) if g2 else 5 + i)) for i in range(64)]
    jsig = [(g1_to_jac if g2 else g2_to_jac)(sigs[i].tobytes(), (7 + i if g2 else (2 + i, 1))) for i in range(min(nmax, 256))]
    for n in sizes:
        msgs = [bytes(packed.buf[int(packed.off[i]):int(packed.off[i + 1])]) for i in range(n)]
        pm = E.PackedMsgs(msgs)
        a, b = np.ascontiguousarray(pks[:n].reshape(-1)), np.ascontiguousarray(sigs[:n].reshape(-1))
        fn = E.g2pubs_verify_batch if g2 else E.g1pubs_verify_batch
        t_host = best(lambda: fn(pm, a, b))
        ok, _ = fn(pm, a, b); assert all(ok)
        t_res = best(lambda: E.verify_batch_dev(pkg, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 0, d_ok.data_ptr(), n))
        prof = bench.profiled(lib, lambda: fn(pm, a, b))
        line = "%s n=%5d  host buffers (affine) %.2f ms   resident %.2f ms   overhead %.2f ms" % (pkg, n, t_host, t_res, t_host - t_res)
        if n <= 256:
            pass
        print(line, {k.replace("k_", ""): round(v[0], 2) for k, v in prof.items()}, flush=True)
