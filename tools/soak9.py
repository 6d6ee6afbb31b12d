"""Concurrency soak of the host side (blsmi 0.6: load-aware layout, the request combiner's two classes, in-memory points): T caller threads issue
verify calls of random size (1 ... 9 000 tuples), package (g2pubs / g1pubs / g1pubs WithDomain), point format (wire / in-memory) and corruption at
the same time; every caller must get exactly its own verdicts (expected values from the oracle, once per base tuple).
python tools/soak9.py [seconds] [threads]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_common import P, RC, g1_to_jac, g2_to_jac  # noqa: E402
from bls_amd import engine as E  # noqa: E402

E.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
xs = P.XORShift(909)
NB = 48
DOM = bytes([7, 0, 0, 0, 0, 0, 0, 1])
sks = [P.rand_fr(xs).to_bytes(32, "big") for _ in range(NB)]
base = {}
for kind in ("g2pubs", "g1pubs", "domain"):
    o = RC.g2pubs if kind == "g2pubs" else RC.g1pubs
    rows = []
    for i, sk in enumerate(sks):
        m = (b"soak9 %d" % i) if kind != "domain" else bytes([i + 1]) * 32
        pk = o.priv_to_pub(sk)
        sg = o.sign(m, sk) if kind != "domain" else o.sign_with_domain(m, sk, DOM)
        bad = i % 5 == 3
        if bad:
            sg = o.sign(b"another", sk) if kind != "domain" else o.sign_with_domain(bytes(32), sk, DOM)
        want = (o.verify(m, pk, sg) if kind != "domain" else o.verify_with_domain(m, pk, sg, DOM))
        assert want == (not bad)
        g1 = kind == "g2pubs"                                              # g2pubs: keys in G2, signatures in G1
        rows.append((m, pk, sg, (g2_to_jac(pk, (5 + i, 3)) if g1 else g1_to_jac(pk, 9 + i)), (g1_to_jac(sg, 4 + i) if g1 else g2_to_jac(sg, (2 + i, 1))), bool(want)))
    base[kind] = rows
SIZES = [1, 1, 3, 7, 64, 300, 1100, 1700, 2600, 4200, 7000, 9000]
stop = time.time() + budget
errs, counts = [], [0] * T


def caller(k):
    rng = np.random.default_rng(1000 + k)
    try:
        while time.time() < stop:
            kind = ("g2pubs", "g1pubs", "domain")[int(rng.integers(3))]
            n = int(SIZES[int(rng.integers(len(SIZES)))])
            jac = bool(rng.integers(2))
            rows = base[kind]
            sel = rng.integers(NB, size=min(n, 96))
            sel = np.resize(sel, n)
            ms = [rows[i][0] for i in sel]
            pk = b"".join(rows[i][3 if jac else 1] for i in sel); sg = b"".join(rows[i][4 if jac else 2] for i in sel)
            want = [rows[i][5] for i in sel]
            if kind == "g2pubs":
                ok = (E.g2pubs_verify_batch_jac if jac else E.g2pubs_verify_batch)(E.PackedMsgs(ms), pk, sg)[0]
            elif kind == "g1pubs":
                ok = (E.g1pubs_verify_batch_jac if jac else E.g1pubs_verify_batch)(E.PackedMsgs(ms), pk, sg)[0]
            else:
                ok = (E.g1pubs_verify_with_domain_batch_jac if jac else E.g1pubs_verify_with_domain_batch)(ms, DOM, pk, sg)
                ok = ok[0] if isinstance(ok, tuple) else ok
            assert [bool(x) for x in ok] == want, (k, kind, n, jac)
            counts[k] += n
    except Exception as e:                                                      # noqa: BLE001
        errs.append(repr(e))


ts = [threading.Thread(target=caller, args=(k,)) for k in range(T)]
t0 = time.time()
for t in ts:
    t.start()
for t in ts:
    t.join()
assert not errs, errs[:3]
print("soak9 ok: %d threads, %d tuples in %.0f s (%.0f verifies/s), GPU_MAX_HW_QUEUES=%s" % (T, sum(counts), time.time() - t0, sum(counts) / (time.time() - t0), os.environ.get("GPU_MAX_HW_QUEUES", "unset")))
