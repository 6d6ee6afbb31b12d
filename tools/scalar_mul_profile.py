"""The three kernels VERDICT r03 item 5 asks about -- k_g1_mul_glv, k_g2_mul_glv_pair (2^20 resident scalar multiplications) and
k_miller1x2_pair (one 2^20-message VerifyAggregate) -- launched a few times each, for rocprofv3 --pmc passes (TCC hit / miss, FETCH_SIZE,
WRITE_SIZE).  Prints wall times so that a PMC pass can be matched to an unprofiled one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from bls_amd import engine as E

E.init(0)
dev = torch.device("cuda", 0)
n = 1 << 20
rng = np.random.default_rng(3)
k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f
d_k = torch.from_numpy(k.reshape(-1)).to(dev)
for grp, pb in (("g1", 96), ("g2", 192)):
    bk = rng.integers(0, 256, size=(4096, 32), dtype=np.uint8); bk[:, 0] &= 0x3f
    bpts, _ = (E.g1_mul_generator_batch if grp == "g1" else E.g2_mul_generator_batch)(bk.reshape(-1), 4096)
    d_p = torch.from_numpy(np.ascontiguousarray(np.tile(bpts, (n // 4096, 1))).reshape(-1)).to(dev)
    d_out = torch.empty(n * pb, dtype=torch.uint8, device=dev); d_inf = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        E.mul_batch_dev(grp, d_p.data_ptr(), d_k.data_ptr(), d_out.data_ptr(), d_inf.data_ptr(), n)
        torch.cuda.synchronize(); print("%s mul 2^20: %.2f ms" % (grp, (time.perf_counter() - t0) * 1e3), flush=True)
    del d_p, d_out
packed, allpk, agg, _ = bench._aggregate_inputs(E, "g2pubs", 0, n)
d_m = torch.from_numpy(packed.buf.copy()).to(dev); d_o = torch.from_numpy(packed.off.view(np.int64).copy()).to(dev); d_pk = torch.from_numpy(allpk).to(dev)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    assert E.verify_aggregate_dev("g2pubs", d_m.data_ptr(), d_o.data_ptr(), d_pk.data_ptr(), agg, n) is True
    torch.cuda.synchronize(); print("g2pubs VerifyAggregate 2^20: %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
