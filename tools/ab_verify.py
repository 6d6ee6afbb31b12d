"""A/B timing of the device-resident verify path for several builds of the library in one process:
   python tools/ab_verify.py lib_a.so lib_b.so ...   (inputs are generated with the in-tree build)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from bls_amd import engine
engine.init(0)
dev = torch.device("cuda", 0)
n = 65536
inputs = {g: bench._verify_inputs(engine, dev, g, n) for g in ("g2pubs", "g1pubs")}
for path in sys.argv[1:]:
    lib = C.CDLL(os.path.abspath(path))
    assert lib.blsmi_init(0) == 0
    for g in ("g2pubs", "g1pubs"):
        d = inputs[g]
        ok = torch.zeros(n, dtype=torch.uint8, device=dev)
        fn = lib.blsmi_g2pubs_verify_batch_dev if g == "g2pubs" else lib.blsmi_g1pubs_verify_batch_dev
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rc = fn(C.c_void_p(d[0].data_ptr()), C.c_void_p(d[1].data_ptr()), C.c_void_p(d[2].data_ptr()), C.c_void_p(d[3].data_ptr()), None, C.c_void_p(ok.data_ptr()), C.c_size_t(n), None)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        assert rc == 0 and bool(ok.all().item())
        print("%s %s: %.3f ms -> %.0f verifies/s" % (os.path.basename(path), g, best * 1e3, n / best))
