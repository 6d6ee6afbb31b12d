"""Round-4 differential soak (round 6: four layouts): the same random tuples through the pairing layouts -- one tuple per wave (k_lat.hip: 15 x 27-bit
limbs), lane ROW (sixteen lanes per tuple, row_body.inc), lane quad and lane pair (14 x 28-bit limbs) -- must give the same Fq12 bits and the same
verdicts; samples go to the oracle.  Two
representations and three lane layouts computing independently find what a handful of KATs cannot (a carry that only a rare limb
pattern produces, a value bound that holds for random inputs only).  Points include the generators, small multiples, r - 1, and points
OUTSIDE the subgroup; verify tuples carry random corruptions.   python tools/soak6.py [seconds]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from bls_amd import engine as E
from oracle import refcpu as RC, pyref as P
E.init(0)
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "20260929")))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
R = P.R_ORDER
t0 = time.time(); rounds = 0; checked = 0


def scalars(n):
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f
    special = [1, 2, 3, R - 1, R - 2, (1 << 64) - 1, 1 << 128, P.BLS_X, P.BLS_X ** 2]
    for i, v in enumerate(special):
        if i < n:
            k[i] = np.frombuffer(int(v % R or 1).to_bytes(32, "big"), dtype=np.uint8)
    return k


def paths():
    yield "wave", 1 << 20, 0, (0, 0)
    yield "row", 1 << 20, 0, (1, 1 << 20)                                # (the g1pubs Verify then also runs its signature side beside the hash: k_miller1s_row / k_miller1m_row)
    yield "quad", 0, 1 << 20, (0, 0)
    yield "pair", 0, 0, (0, 0)


from test_gpu_round3 import _torsion_points
T1, T2 = _torsion_points()
try:
    while time.time() - t0 < budget:
        n = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 100, 257, 1000, 2049, 3000]))
        ka, kb = scalars(n), scalars(n)
        g1, _ = E.g1_mul_generator_batch(ka.reshape(-1), n); g2, _ = E.g2_mul_generator_batch(kb.reshape(-1), n)
        g1 = g1.copy(); g2 = g2.copy()
        for j in range(min(3, n)):                                      # a few points outside the subgroup
            i = int(rng.integers(0, n))
            if rng.integers(0, 2):
                g1[i] = np.frombuffer(T1[j % len(T1)], dtype=np.uint8)
            else:
                g2[i] = np.frombuffer(T2[j % len(T2)], dtype=np.uint8)
        outs = {}
        for name, lat, quad, row in paths():
            E.set_latency_threshold(lat); E.set_quad_threshold(quad); E.set_row_threshold(*row)
            outs[name] = E.pairing_batch(g1.reshape(-1), g2.reshape(-1), n)
        assert np.array_equal(outs["wave"], outs["quad"]) and np.array_equal(outs["quad"], outs["pair"]) and np.array_equal(outs["row"], outs["pair"]), ("pairing", n)
        for i in rng.integers(0, n, size=min(n, 3)):
            assert np.array_equal(outs["pair"][i], RC.pairing_batch(g1[i].tobytes(), g2[i].tobytes(), 1)[0]), ("pairing oracle", n, int(i))
            checked += 1
        # verifies of both packages with random corruptions
        for group, O in (("g2pubs", RC.g2pubs), ("g1pubs", RC.g1pubs)):
            nk = 8
            sks = [hashlib.sha256(b"soak6-%d-%d" % (rounds, i)).digest()[:31].rjust(32, b"\0") for i in range(nk)]
            msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 70)), dtype=np.uint8)) for _ in range(n)]
            if group == "g2pubs":
                pks, _ = E.g2_mul_generator_batch(b"".join(sks), nk); h = E.hash_g1_batch(msgs)
                sig, _ = E.g1_mul_batch(h.reshape(-1), b"".join(sks[i % nk] for i in range(n)), n)
            else:
                pks, _ = E.g1_mul_generator_batch(b"".join(sks), nk); h = E.hash_g2_batch(msgs)
                sig, _ = E.g2_mul_batch(h.reshape(-1), b"".join(sks[i % nk] for i in range(n)), n)
            allpk = np.stack([pks[i % nk] for i in range(n)]).copy(); sig = sig.copy()
            expect = np.ones(n, dtype=bool)
            for i in rng.integers(0, n, size=max(1, n // 7)):
                kind = int(rng.integers(0, 3)); i = int(i)
                if kind == 0:
                    allpk[i] = pks[(i + 1) % nk]
                elif kind == 1:
                    msgs[i] = msgs[i] + b"!"
                else:
                    sig[i] = sig[(i + 1) % n] if n > 1 else sig[i]
                    if n == 1:
                        continue
                expect[i] = False
            fn = E.g2pubs_verify_batch if group == "g2pubs" else E.g1pubs_verify_batch
            oks = {}
            for name, lat, quad, row in paths():
                E.set_latency_threshold(lat); E.set_quad_threshold(quad); E.set_row_threshold(*row)
                E.set_option("row_side", int(rng.integers(0, 2)))       # (either form of the row layout's g1pubs Verify)
                E.set_option("row_side_g2pubs", int(rng.integers(0, 2))); E.set_option("row_side_piece", int(rng.choice([0, 0, 7, 1000])))   # (... of g2pubs; the side kernel in pieces)
                E.set_option("swu_row_max", int(rng.choice([0, 4096, 1 << 20])))                                                      # (the maps a lane or a row of sixteen each)
                tail = int(rng.integers(0, 4))                                                                                          # HashG2's tail: as shipped / sixteen lanes / four lanes
                E.set_option("hash_row_min", (2048, 1, 1, 1)[tail]); E.set_option("hash_row_max", (4096, 1 << 20, 0, 0)[tail]); E.set_option("hash_quad_min", (4097, 1, 1, 1)[tail]); E.set_option("hash_quad_max", (16384, 0, 1 << 20, 0)[tail])
                E.set_option("hash_oct_min", (2048, 1, 1, 1)[tail]); E.set_option("hash_oct_max", (7168, 0, 0, 1 << 20)[tail])   # ... / eight lanes for every count
                E.set_option("hash_g1_quad_min", int(rng.choice([1, 1280])))
                oks[name], _ = fn(msgs, allpk.reshape(-1), sig.reshape(-1))
            assert np.array_equal(oks["wave"], oks["quad"]) and np.array_equal(oks["quad"], oks["pair"]) and np.array_equal(oks["row"], oks["pair"]), (group, n)
            bad = np.nonzero(oks["pair"] != expect)[0]
            for i in bad:                                               # a "corruption" can coincide with the truth (same key twice): ask the oracle
                assert O.verify(msgs[i], allpk[i].tobytes(), sig[i].tobytes()) == bool(oks["pair"][i]), (group, n, int(i))
            for i in rng.integers(0, n, size=2):
                assert O.verify(msgs[int(i)], allpk[int(i)].tobytes(), sig[int(i)].tobytes()) == bool(oks["pair"][int(i)]); checked += 1
        rounds += 1
finally:
    E.set_latency_threshold(8192); E.set_quad_threshold(16384); E.set_row_threshold(*E.ROW_DEFAULT); E.set_option("row_side", 1)
    for k, v in (("row_side_g2pubs", 1), ("row_side_piece", 0), ("swu_row_max", 4096), ("hash_row_min", 2048), ("hash_row_max", 4096), ("hash_quad_min", 4097), ("hash_quad_max", 16384), ("hash_oct_min", 2048), ("hash_oct_max", 7168), ("hash_g1_quad_min", 1280)):
        E.set_option(k, v)
print("soak6: %d rounds, %d oracle samples, %.0f s: four layouts / two limb representations agree" % (rounds, checked, time.time() - t0))
