"""Where should BLSMI_SIG_SIDE_MAX sit?  One g2pubs / g1pubs verify call of n tuples (host buffers), wall time and kernel times, with the
signature side on the side stream (BLSMI_SIG_SIDE_MAX = 1 << 20) and without (0): run once per setting (the variable is read at load)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bls_amd import engine as E, _native
E.init(0)
lib = _native.load()
for group in ("g2pubs", "g1pubs"):
    packed, pks, sigs = bench._verify_tuples(E, group, 4096, tag=3)
    fn = E.g2pubs_verify_batch if group == "g2pubs" else E.g1pubs_verify_batch
    pb, sb = pks.shape[1], sigs.shape[1]
    for n in ([int(x) for x in sys.argv[1:]] or [1, 16, 64, 128, 256, 512, 1000, 1024, 2048, 4096]):
        msgs = [bytes(packed.buf[int(packed.off[i]):int(packed.off[i + 1])]) for i in range(n)]
        pm = E.PackedMsgs(msgs)
        p = pks[:n].reshape(-1); s = sigs[:n].reshape(-1)
        f = lambda: fn(pm, p, s)
        ok, _ = f(); assert bool(ok.all())
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
        prof = bench.profiled(lib, f)
        print("%s n=%5d side_max=%s  %.3f ms  %s" % (group, n, os.environ.get("BLSMI_SIG_SIDE_MAX", "default"), best * 1e3, {k: round(v[0], 2) for k, v in prof.items() if not k.startswith("(")}), flush=True)
