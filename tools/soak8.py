"""Differential soak of the in-memory-point boundary (blsmi 0.6): random batches of both packages, random batch sizes across the three layouts,
random representatives (z random, z = 1, z = 0, a limb image >= q), random corruptions -- the *_jac entry points must return the verdicts of the
affine entry points on ToAffine of the same points, small cases are also put to the oracle.  python tools/soak8.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_common import P, RC, g1_to_jac, g2_to_jac  # noqa: E402
from bls_amd import engine as E  # noqa: E402

E.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "80808")))
xs = P.XORShift(808)
NK = 48
sks = [P.rand_fr(xs).to_bytes(32, "big") for _ in range(NK)]
pk = {"g2pubs": [RC.g2pubs.priv_to_pub(s) for s in sks], "g1pubs": [RC.g1pubs.priv_to_pub(s) for s in sks]}
MS = [b"soak8 message %d" % i for i in range(64)]
sg = {g: {} for g in pk}


def sig(g, i, j):
    if (i, j) not in sg[g]:
        sg[g][(i, j)] = (RC.g2pubs if g == "g2pubs" else RC.g1pubs).sign(MS[j], sks[i])
    return sg[g][(i, j)]


def rz1():
    return int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % (P.Q - 1) + 1


def rep(group_is_g1, wire, mode):
    """mode 0: random z; 1: z = 1; 2: z = 0 (infinity); 3: z + q where it fits in 384 bits (reads as 0: infinity)"""
    if group_is_g1:
        j = g1_to_jac(wire, 1 if mode == 1 else 0 if mode == 2 else rz1())
    else:
        j = g2_to_jac(wire, (1, 0) if mode == 1 else (0, 0) if mode == 2 else (rz1(), rz1() - 1))
    if mode == 3:
        a = np.frombuffer(j, dtype=np.uint64).copy()
        nc = 1 if group_is_g1 else 2
        for e in range(nc):
            v = P.from_limbs64(a[6 * (2 * nc + e):6 * (2 * nc + e) + 6]) + P.Q
            if v >= 1 << 384:
                return j, 0                                                  # no room: stays an ordinary point
            a[6 * (2 * nc + e):6 * (2 * nc + e) + 6] = np.array(P.limbs64(v), dtype=np.uint64)
        return a.tobytes(), 3
    return j, mode


t0 = time.time()
rounds = tuples = oracle = 0
sizes = [1, 2, 3, 17, 64, 65, 200, 1500, 6000, 9000, 20000]
while time.time() - t0 < budget:
    g = "g2pubs" if rng.integers(2) else "g1pubs"
    n = int(sizes[int(rng.integers(len(sizes)))])
    base = min(n, 96)
    ki = rng.integers(NK, size=base); mi = rng.integers(len(MS), size=base)
    msgs, pks, sigs, jp, js, flags, want_inf = [], [], [], [], [], [], []
    for t in range(base):
        k, m = int(ki[t]), int(mi[t])
        corrupt = int(rng.integers(8))
        p = pk[g][k]; s = sig(g, k, m); msg = MS[m]
        if corrupt == 0:
            p = pk[g][(k + 1) % NK]
        elif corrupt == 1:
            msg = MS[(m + 1) % len(MS)]
        elif corrupt == 2:
            s = sig(g, (k + 1) % NK, m)
        pm = int(rng.choice([0, 0, 0, 1, 1, 2, 3])); sm = int(rng.choice([0, 0, 0, 1, 1, 2, 3]))
        a, pm = rep(g == "g1pubs", p, pm)
        b, sm = rep(g == "g2pubs", s, sm)
        msgs.append(msg); pks.append(p); sigs.append(s); jp.append(a); js.append(b)
        flags.append((1 if pm >= 2 else 0) | (2 if sm >= 2 else 0))
    reps = (n + base - 1) // base
    sel = (list(range(base)) * reps)[:n]
    M = E.PackedMsgs([msgs[i] for i in sel])
    A = b"".join(pks[i] for i in sel); B = b"".join(sigs[i] for i in sel)
    JA = b"".join(jp[i] for i in sel); JB = b"".join(js[i] for i in sel)
    F = np.array([flags[i] for i in sel], dtype=np.uint8)
    if g == "g2pubs":
        aff, _ = E.g2pubs_verify_batch(M, A, B, F); jac, bm = E.g2pubs_verify_batch_jac(M, JA, JB)
    else:
        aff, _ = E.g1pubs_verify_batch(M, A, B, F); jac, bm = E.g1pubs_verify_batch_jac(M, JA, JB)
    assert np.array_equal(aff, jac), (g, n, np.nonzero(aff != jac)[0][:5])
    assert bytes(bm) == bytes(np.packbits(jac, bitorder="little"))
    if rounds % 5 == 0:
        o = RC.g2pubs if g == "g2pubs" else RC.g1pubs
        for t in range(min(base, 12)):
            w = bool(o.verify(msgs[t], pks[t], sigs[t])) and flags[t] == 0
            assert bool(jac[t]) == w, (g, n, t)
            oracle += 1
    rounds += 1; tuples += n
print("soak8 ok: %d rounds, %d tuples, %d oracle comparisons, %.0f s" % (rounds, tuples, oracle, time.time() - t0))
