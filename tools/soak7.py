"""Round-4 soak of VerifyAggregate (g2pubs and g1pubs): random sizes on every path -- latency programs with the tail in two pieces, lane quads,
lane pairs, and from 65 536 messages the g2pubs path that pairs the hash points before their cofactor clearing and raises the Miller product to
1 - x -- with random corruptions (wrong key, wrong message, tampered aggregate, a duplicated message, a key at infinity); the verdict must be the
expected one, must not depend on BLSMI_AGG_COFACTOR_POW, and small cases go to the oracle.   python tools/soak7.py [seconds]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from bls_amd import engine as E
from oracle import refcpu as RC
E.init(0)
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "70707")))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nk = 97
sk = [hashlib.sha256(b"soak7-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk)]
PK = {"g2pubs": E.g2_mul_generator_batch(b"".join(sk), nk)[0], "g1pubs": E.g1_mul_generator_batch(b"".join(sk), nk)[0]}
t0 = time.time(); rounds = 0; oracle = 0; big = 0
sizes = [1, 2, 3, 31, 64, 128, 200, 1000, 4095, 4097, 6000, 12000, 20000, 40000, 65535, 65536, 65537, 70001, 131072]
while time.time() - t0 < budget:
    group = "g2pubs" if rng.random() < 0.65 else "g1pubs"
    n = int(rng.choice(sizes[:-4] if group == "g1pubs" and rng.random() < 0.8 else sizes))
    tag = int(rng.integers(0, 1 << 30))
    msgs = [b"%d:%d" % (tag, i) + b"x" * int(i % 5) for i in range(n)]
    sks = (b"".join(sk) * (n // nk + 1))[:32 * n]
    sign = E.g2pubs_sign_batch if group == "g2pubs" else E.g1pubs_sign_batch
    summ = E.g1_sum if group == "g2pubs" else E.g2_sum
    pkb = 192 if group == "g2pubs" else 96
    sigs, _ = sign(msgs, sks)
    agg = summ(sigs.reshape(-1), n)
    pks = np.ascontiguousarray(np.tile(PK[group], (n // nk + 1, 1))[:n])
    kind = int(rng.integers(0, 6))
    want = True
    m2, p2, a2 = msgs, pks, agg
    j = int(rng.integers(0, n))
    if kind == 1:
        p2 = pks.copy(); p2[j] = PK[group][(j + 1) % nk]; want = False
    elif kind == 2:
        m2 = list(msgs); m2[j] = b"tampered"; want = False
    elif kind == 3 and n > 1:
        a2 = summ(sigs[:n - 1].reshape(-1), n - 1); want = False
    elif kind == 4 and n > 1:
        m2 = list(msgs); m2[j] = msgs[(j + 1) % n]; want = False                 # a duplicate (the reference rejects, bls.go:245-261)
    elif kind == 5:
        p2 = pks.copy(); p2[j] = 0; want = False                                  # a key at infinity (all-zero record)
    verify = E.g2pubs_verify_aggregate if group == "g2pubs" else E.g1pubs_verify_aggregate
    got = []
    for mode in (1, 0):
        E.set_option("agg_cofactor_pow", mode)
        got.append(verify(m2, p2.reshape(-1), a2))
    assert got[0] == got[1] == want, (group, n, kind, got, want)
    if n <= 64 and kind != 5:
        o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
        assert o.verify_aggregate(a2, [p2[i].tobytes() for i in range(n)], m2) == want, (group, n, kind)
        oracle += 1
    big += n >= 65536
    rounds += 1
E.set_option("agg_cofactor_pow", 1)
print("soak7 ok: %d rounds (%d at 65 536 messages or more), %d oracle comparisons, %.0f s" % (rounds, big, oracle, time.time() - t0))
