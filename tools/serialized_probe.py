"""Deserialize + Verify throughput: 65 536 (message, compressed key, compressed signature) tuples through
blsmi_g{1,2}pubs_verify_serialized_batch (host buffers), with the library's per-kernel log.  python tools/serialized_probe.py [n]"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bls_amd import engine as eng  # noqa: E402

eng.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nk = 4096
sk = b"".join(hashlib.sha256(b"sp-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
skb = np.frombuffer(sk, dtype=np.uint8).reshape(nk, 32)
msgs = [hashlib.sha256(b"spm%d" % i).digest() for i in range(n)]
idx = np.arange(n) % nk
pm = eng.PackedMsgs(msgs)
for group in ("g2pubs", "g1pubs"):
    if group == "g2pubs":
        pks, _ = eng.g2_mul_generator_batch(sk, nk); h = eng.hash_g1_batch(pm)
        sigs, _ = eng.g1_mul_batch(h.reshape(-1), skb[idx].reshape(-1), n)
        pkc = eng.g2_compress_batch(pks[idx].reshape(-1), n); sgc = eng.g1_compress_batch(sigs.reshape(-1), n)
    else:
        pks, _ = eng.g1_mul_generator_batch(sk, nk); h = eng.hash_g2_batch(pm)
        sigs, _ = eng.g2_mul_batch(h.reshape(-1), skb[idx].reshape(-1), n)
        pkc = eng.g1_compress_batch(pks[idx].reshape(-1), n); sgc = eng.g2_compress_batch(sigs.reshape(-1), n)
    for check in (True, False):
        fn = lambda: eng.verify_serialized_batch(group, pm, pkc.reshape(-1), sgc.reshape(-1), check_subgroup=check)
        ok, ep, es = fn()
        assert ok.all() and not ep.any() and not es.any()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
        prof = bench.profiled(eng._lib(), fn)
        print("%s n=%d subgroup_check=%s: %.2f ms (%.2f M/s, host buffers)  %s" % (group, n, check, best * 1e3, n / best / 1e6,
              {k: round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8]}))
