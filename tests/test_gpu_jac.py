"""-m gpu, round 5: the reference's in-memory points at the C ABI (blsmi 0.6, the *_jac entry points).

A Go caller holds *bls.G1Projective / *bls.G2Projective (g2pubs/bls.go:13-15, 53-55): Jacobian coordinates, Montgomery(2^384) limbs.
Every test hands the library RANDOM representatives (z != 1) of the same tuples and requires (i) the verdicts / Fq12 bits of the affine
entry points, (ii) the oracle's verdicts on RC.g?_jac_to_affine_bytes of the same limbs -- on the latency path, the quad kernels and
the lane-pair kernels."""
import threading

import numpy as np
import pytest

from gpu_common import P, RC, g1_to_jac, g2_to_jac, jac1, jac2, mont, rand_g1, rand_g2, rand_z1, rand_z2, sk_bytes

pytestmark = pytest.mark.gpu

PATHS = [(8192, 16384), (0, 16384), (0, 0), (-1, 16384)]      # (latency threshold, quad threshold): one tuple per wave / per lane quad / per lane pair; -1: per lane ROW (round 6)


@pytest.fixture(scope="module")
def eng():
    from bls_amd import engine
    engine.init(0)
    yield engine
    engine.set_latency_threshold(8192); engine.set_quad_threshold(16384); engine.set_row_threshold(*engine.ROW_DEFAULT)


def _set_path(eng, lat, quad):
    """lat == -1: the lane-row kernels whatever the size; otherwise the row layout off and the thresholds as given"""
    eng.set_row_threshold(*((1, 1 << 20) if lat < 0 else (0, 0)))
    eng.set_latency_threshold(8192 if lat < 0 else lat); eng.set_quad_threshold(quad)


def _u64(b):
    return np.frombuffer(b, dtype=np.uint64)


def _oracle_affine1(j):
    return RC.g1_jac_to_affine_bytes(_u64(j))


def _oracle_affine2(j):
    return RC.g2_jac_to_affine_bytes(_u64(j))


def _add_q(rec, coord):
    """the same record with coordinate `coord` (index of a 6-limb FQ) replaced by its image + q: not below q, still 384 bits"""
    a = _u64(rec).copy()
    v = P.from_limbs64(a[6 * coord:6 * coord + 6]) + P.Q
    assert v < 1 << 384
    a[6 * coord:6 * coord + 6] = np.array(P.limbs64(v), dtype=np.uint64)
    return a.tobytes()


def _zero_coord(rec, coord):
    a = _u64(rec).copy()
    a[6 * coord:6 * coord + 6] = 0
    return a.tobytes()


def test_to_affine_matches_the_oracle(eng):
    """ToAffine + SerializeBytes on the device against the oracle's (g1.go:322-340 + 157-167, g2.go:365-386 + 172-186): random z, z = 1
    (the wave-uniform shortcut: 64 and 128 such points; and mixed into a wave of others), z = 0, ragged counts."""
    xs = P.XORShift(5101)
    w1 = [rand_g1(xs) for _ in range(9)]; w2 = [rand_g2(xs) for _ in range(9)]
    for n in (1, 63, 64, 129):
        j1 = [jac1(xs, w1[i % 9]) for i in range(n)]; j2 = [jac2(xs, w2[i % 9]) for i in range(n)]
        if n > 3:
            j1[1] = g1_to_jac(w1[1]); j2[1] = g2_to_jac(w2[1])                   # z = 1 inside a wave of others
            j1[2] = g1_to_jac(w1[2], 0); j2[2] = g2_to_jac(w2[2], (0, 0))       # z = 0: infinity, whatever x and y hold
            j1[3] = g1_to_jac(None); j2[3] = g2_to_jac(None)                    # the reference's G?ProjectiveZero
        o1, i1 = eng.g1_jac_to_affine_batch(b"".join(j1), n)
        o2, i2 = eng.g2_jac_to_affine_batch(b"".join(j2), n)
        for t in range(n):
            want1, want2 = _oracle_affine1(j1[t]), _oracle_affine2(j2[t])
            assert bool(i1[t]) == (want1 is None) and bool(i2[t]) == (want2 is None), (n, t)
            assert o1[96 * t:96 * t + 96] == (want1 or bytes(96)), (n, t)
            assert o2[192 * t:192 * t + 192] == (want2 or bytes(192)), (n, t)
    for n in (64, 128, 130):                                                   # whole waves of z = 1: no inversion runs
        j1 = [g1_to_jac(w1[i % 9]) for i in range(n)]; j2 = [g2_to_jac(w2[i % 9]) for i in range(n)]
        o1, i1 = eng.g1_jac_to_affine_batch(b"".join(j1), n)
        o2, i2 = eng.g2_jac_to_affine_batch(b"".join(j2), n)
        assert o1 == b"".join(w1[i % 9] for i in range(n)) and o2 == b"".join(w2[i % 9] for i in range(n)) and not i1.any() and not i2.any()
    assert eng.g1_jac_to_affine_batch(b"", 0)[0] == b""


def test_limb_images_not_below_q_read_as_zero(eng):
    """A coordinate >= q cannot come out of the reference's arithmetic (fq.go:37-45); FQReprToFQ turns an invalid repr into 0
    (fq.go:49-56) and so does the boundary: the record behaves like the one with that coordinate zeroed -- z: the point at infinity."""
    xs = P.XORShift(5102)
    w1, w2 = rand_g1(xs), rand_g2(xs)
    j1, j2 = jac1(xs, w1), jac2(xs, w2)
    small = None
    for _ in range(64):                                                        # a representative whose limb images leave room for + q
        j1, j2 = jac1(xs, w1), jac2(xs, w2)
        if all(P.from_limbs64(_u64(j1)[6 * c:6 * c + 6]) + P.Q < 1 << 384 for c in range(3)) and all(P.from_limbs64(_u64(j2)[6 * c:6 * c + 6]) + P.Q < 1 << 384 for c in range(6)):
            small = True
            break
    assert small
    recs1 = [_add_q(j1, c) for c in range(3)]; want1 = [_zero_coord(j1, c) for c in range(3)]
    recs2 = [_add_q(j2, c) for c in range(6)]; want2 = [_zero_coord(j2, c) for c in range(6)]
    o1, i1 = eng.g1_jac_to_affine_batch(b"".join(recs1), 3)
    e1, f1 = eng.g1_jac_to_affine_batch(b"".join(want1), 3)
    o2, i2 = eng.g2_jac_to_affine_batch(b"".join(recs2), 6)
    e2, f2 = eng.g2_jac_to_affine_batch(b"".join(want2), 6)
    assert o1 == e1 and list(i1) == list(f1) == [False, False, True]
    assert o2 == e2 and list(i2) == list(f2) == [False] * 4 + [False, False]   # one half of z zeroed: z != 0 still
    for c in range(3):
        w = _oracle_affine1(want1[c])
        assert o1[96 * c:96 * c + 96] == (w or bytes(96))
    # both halves of a G2 z out of range: infinity
    both = _add_q(_add_q(j2, 4), 5)
    o, i = eng.g2_jac_to_affine_batch(both, 1)
    assert i[0] and o == bytes(192)
    # ... and a verify over such a key is False, not undefined (the host-side screen of an aggregate's signature included)
    msg = b"m"
    sk = sk_bytes(xs)
    pk, sig = RC.g2pubs.priv_to_pub(sk), RC.g2pubs.sign(msg, sk)
    ok, _ = eng.g2pubs_verify_batch_jac([msg, msg], g2_to_jac(pk) + _add_q(_add_q(g2_to_jac(pk), 4), 5), g1_to_jac(sig) * 2)
    assert list(ok) == [True, False]
    assert eng.g2pubs_verify_aggregate_jac([msg], g2_to_jac(pk), _add_q(g1_to_jac(sig), 2)) is False


@pytest.mark.parametrize("lat,quad", PATHS)
def test_pairing_of_in_memory_points(eng, lat, quad):
    """bls.Pairing(p *G1Projective, q *G2Projective) (pairing.go:132-136): Fq12 bits identical to the affine entry point and the oracle"""
    _set_path(eng, lat, quad)
    xs = P.XORShift(5103)
    for n in (1, 70):
        w1 = [rand_g1(xs) for _ in range(min(n, 5))]; w2 = [rand_g2(xs) for _ in range(min(n, 5))]
        a = b"".join(w1[i % len(w1)] for i in range(n)); b = b"".join(w2[i % len(w2)] for i in range(n))
        ja = b"".join(jac1(xs, w1[i % len(w1)]) for i in range(n)); jb = b"".join(jac2(xs, w2[i % len(w2)]) for i in range(n))
        got = eng.pairing_batch_jac(ja, jb, n)
        assert np.array_equal(got, eng.pairing_batch(a, b, n))
        assert np.array_equal(got[:min(n, 5)], RC.pairing_batch(a[:96 * min(n, 5)], b[:192 * min(n, 5)], min(n, 5)))


def _tuples(group, n, xs, domain=None):
    """n (message, key, signature) tuples of one package with every 5th corrupted (wrong message / wrong key / negated-by-swap signature);
    returns wire records and the oracle's verdicts"""
    R = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    sks = [sk_bytes(xs) for _ in range(n)]
    if domain is None:
        msgs = [b"Hello world! 16 characters %d" % i for i in range(n)]
        sign = R.sign
    else:
        msgs = [RC.sha256(b"%d" % i) for i in range(n)]
        sign = lambda m, sk: R.sign_with_domain(m, sk, domain)                  # noqa: E731
    pks = [R.priv_to_pub(sk) for sk in sks]
    sigs = [sign(m, sk) for m, sk in zip(msgs, sks)]
    for i in range(0, n, 5):
        kind = (i // 5) % 3
        if kind == 0:
            msgs[i] = msgs[i] + b"!" if domain is None else RC.sha256(msgs[i])
        elif kind == 1:
            pks[i] = pks[(i + 1) % n] if n > 1 else R.priv_to_pub(sk_bytes(xs))
        else:
            sigs[i] = sigs[(i + 1) % n] if n > 1 else sign(msgs[i], sk_bytes(xs))
    return msgs, pks, sigs


@pytest.mark.parametrize("lat,quad", PATHS)
@pytest.mark.parametrize("group", ["g2pubs", "g1pubs", "g1pubs_domain"])
def test_verify_batch_of_in_memory_points(eng, group, lat, quad):
    """Verify x n (g2pubs/bls.go:159-162, g1pubs/bls.go:165-174) over random representatives: verdicts equal to the affine entry point's and
    to the oracle's; a key or signature with z = 0 gives False (the reference panics in MillerLoop)."""
    _set_path(eng, lat, quad)
    xs = P.XORShift(5104 + len(group))
    domain = b"\x01\x02\x03\x04\x05\x06\x07\x08" if group.endswith("domain") else None
    pkg = "g2pubs" if group == "g2pubs" else "g1pubs"
    pj, sj = (jac2, jac1) if pkg == "g2pubs" else (jac1, jac2)
    pz, sz = ((lambda w: g2_to_jac(w, (0, 0))), (lambda w: g1_to_jac(w, 0))) if pkg == "g2pubs" else ((lambda w: g1_to_jac(w, 0)), (lambda w: g2_to_jac(w, (0, 0))))
    R = RC.g2pubs if pkg == "g2pubs" else RC.g1pubs
    for n in (1, 3, 50, 97):                                                   # 1, 3: the signature side on its own stream (verify_sig_side_start)
        msgs, pks, sigs = _tuples(pkg, n, xs, domain)
        jp = [pj(xs, w) for w in pks]; js = [sj(xs, w) for w in sigs]
        inf = np.zeros(n, np.uint8)
        if n >= 50:
            jp[7] = pz(pks[7]); inf[7] |= 1
            js[11] = sz(sigs[11]); inf[11] |= 2
        if domain is None:
            want = [R.verify(m, p, s) for m, p, s in zip(msgs, pks, sigs)]
            aff, _ = (eng.g2pubs_verify_batch if pkg == "g2pubs" else eng.g1pubs_verify_batch)(msgs, b"".join(pks), b"".join(sigs), inf)
            got, bitmap = (eng.g2pubs_verify_batch_jac if pkg == "g2pubs" else eng.g1pubs_verify_batch_jac)(msgs, b"".join(jp), b"".join(js))
            assert bytes(bitmap) == bytes(np.packbits(got, bitorder="little"))
        else:
            want = [R.verify_with_domain(m, p, s, domain) for m, p, s in zip(msgs, pks, sigs)]
            aff = eng.g1pubs_verify_with_domain_batch(msgs, domain, b"".join(pks), b"".join(sigs), inf)
            got = eng.g1pubs_verify_with_domain_batch_jac(msgs, domain, b"".join(jp), b"".join(js))
        want = [w and not inf[i] for i, w in enumerate(want)]
        assert list(got) == list(aff) == want, (group, n)
        assert any(want) and not all(want) or n == 1


@pytest.mark.parametrize("lat,quad", PATHS[:2])
@pytest.mark.parametrize("group", ["g2pubs", "g1pubs", "g1pubs_domain"])
def test_verify_aggregate_of_in_memory_points(eng, group, lat, quad):
    """Signature.VerifyAggregate (g2pubs/bls.go:240-270, g1pubs/bls.go:252-282, :300-311): true aggregate, a wrong key, a duplicate
    message, a key with z = 0, an aggregate signature with z = 0, n = 0"""
    _set_path(eng, lat, quad)
    xs = P.XORShift(5110 + len(group))
    domain = b"\x09\x08\x07\x06\x05\x04\x03\x02" if group.endswith("domain") else None
    pkg = "g2pubs" if group == "g2pubs" else "g1pubs"
    R = RC.g2pubs if pkg == "g2pubs" else RC.g1pubs
    pj, sj, g_sum = (jac2, jac1, RC.g1_sum) if pkg == "g2pubs" else (jac1, jac2, RC.g2_sum)
    pz = (lambda w: g2_to_jac(w, (0, 0))) if pkg == "g2pubs" else (lambda w: g1_to_jac(w, 0))
    sz = (lambda w: g1_to_jac(w, 0)) if pkg == "g2pubs" else (lambda w: g2_to_jac(w, (0, 0)))
    if domain is None:
        agg_j = eng.g2pubs_verify_aggregate_jac if pkg == "g2pubs" else eng.g1pubs_verify_aggregate_jac
        agg_a = eng.g2pubs_verify_aggregate if pkg == "g2pubs" else eng.g1pubs_verify_aggregate
        agg_o = R.verify_aggregate
    else:
        agg_j = lambda m, p, s: eng.g1pubs_verify_aggregate_with_domain_jac(m, domain, p, s)          # noqa: E731
        agg_a = lambda m, p, s: eng.g1pubs_verify_aggregate_with_domain(m, domain, p, s)              # noqa: E731
        agg_o = lambda s, p, m: R.verify_aggregate_with_domain(s, p, m, domain)                       # noqa: E731
    for n in (1, 6, 33):
        sks = [sk_bytes(xs) for _ in range(n)]
        msgs = [RC.sha256(b"agg %d" % i) for i in range(n)]
        pks = [R.priv_to_pub(sk) for sk in sks]
        sigs = [(R.sign(m, sk) if domain is None else R.sign_with_domain(m, sk, domain)) for m, sk in zip(msgs, sks)]
        agg = g_sum(b"".join(sigs), n)
        cases = [("true", msgs, pks, agg)]
        if n > 1:
            cases.append(("wrong key", msgs, [pks[1]] + pks[1:], agg))
            if domain is None:
                cases.append(("duplicate", [msgs[1]] + msgs[1:], pks, agg))
        for name, m, p, s in cases:
            want = agg_o(s, p, m)
            assert agg_a(m, b"".join(p), s) == want, (group, n, name)
            assert agg_j(m, b"".join(pj(xs, w) for w in p), sj(xs, s)) == want, (group, n, name)
            assert want == (name == "true")
        jp = [pj(xs, w) for w in pks]
        assert agg_j(msgs, b"".join(jp), sz(agg)) is False                      # aggregate signature at infinity
        jp[n // 2] = pz(pks[n // 2])
        assert agg_j(msgs, b"".join(jp), sj(xs, agg)) is False                  # a key at infinity
    s1 = (rand_g1 if pkg == "g2pubs" else rand_g2)(xs)
    assert agg_j([], b"", sj(xs, s1)) == agg_a([], b"", s1) is False            # n = 0: e(sig, g) == 1 only for sig = infinity


@pytest.mark.parametrize("group", ["g2pubs", "g1pubs", "g1pubs_domain"])
def test_verify_aggregate_common_of_in_memory_points(eng, group):
    """Signature.VerifyAggregateCommon(+WithDomain) (g2pubs/bls.go:275-278, g1pubs/bls.go:287-297): the keys are summed as Jacobian points
    (k_g?_sum0_jac above 16 384 keys, the level programs below), then one Verify"""
    xs = P.XORShift(5120 + len(group))
    domain = b"\x11\x22\x33\x44\x55\x66\x77\x88" if group.endswith("domain") else None
    pkg = "g2pubs" if group == "g2pubs" else "g1pubs"
    R = RC.g2pubs if pkg == "g2pubs" else RC.g1pubs
    pj, sj, g_sum = (jac2, jac1, RC.g1_sum) if pkg == "g2pubs" else (jac1, jac2, RC.g2_sum)
    msg = RC.sha256(b"common") if domain else b"one message for all"
    base = 12
    sks = [sk_bytes(xs) for _ in range(base)]
    pks = [R.priv_to_pub(sk) for sk in sks]
    sigs = [(R.sign(msg, sk) if domain is None else R.sign_with_domain(msg, sk, domain)) for sk in sks]
    if domain is None:
        com_j = eng.g2pubs_verify_aggregate_common_jac if pkg == "g2pubs" else eng.g1pubs_verify_aggregate_common_jac
        com_a = eng.g2pubs_verify_aggregate_common if pkg == "g2pubs" else eng.g1pubs_verify_aggregate_common
    else:
        com_j = lambda m, p, s, n: eng.g1pubs_verify_aggregate_common_with_domain_jac(m, domain, p, s, n)   # noqa: E731
        com_a = lambda m, p, s, n: eng.g1pubs_verify_aggregate_common_with_domain(m, domain, p, s, n)       # noqa: E731
    for n in (1, 5, 12):
        agg = g_sum(b"".join(sigs[:n]), n)
        assert com_a(msg, b"".join(pks[:n]), agg, n) is True
        assert com_j(msg, b"".join(pj(xs, w) for w in pks[:n]), sj(xs, agg), n) is True
        assert com_j(msg, b"".join(pj(xs, w) for w in pks[:n]), sj(xs, sigs[0] if n > 1 else sigs[1]), n) is False
    # many keys (the throughput sum kernels): the 12 keys repeated, so the aggregate signature is reps * (sum of the 12)
    for n in (12 * 1500, 12 * 11000):                                          # 18 000: one context; 132 000: above COMMON_ONE_LEASE_MAX (the host-split sum)
        reps = n // 12
        sig12 = g_sum(b"".join(sigs), 12)
        agg = (RC.g1_mul if pkg == "g2pubs" else RC.g2_mul)(sig12, reps.to_bytes(32, "big"))
        jrec = [pj(xs, w) for w in pks] + [pj(xs, w) for w in pks]            # two representatives of every key
        jp = b"".join(jrec) * (reps // 2)
        assert com_j(msg, jp, sj(xs, agg), n) is True
        assert com_a(msg, b"".join(pks) * reps, agg, n) is True
        assert com_j(msg, jp, sj(xs, sig12), n) is False
    assert com_j(msg, b"", sj(xs, sigs[0]), 0) is False                         # the empty sum is the point at infinity


def test_point_sums_of_in_memory_points(eng):
    """AggregateSignatures / AggregatePublicKeys (g2pubs/bls.go:165-192) with the sum handed back as an in-memory point (z = 1)"""
    xs = P.XORShift(5130)
    w1 = [rand_g1(xs) for _ in range(10)]; w2 = [rand_g2(xs) for _ in range(10)]
    one = np.concatenate([mont(1)]).tobytes()
    for n in (1, 2, 3, 10, 200, 20000):
        a = [w1[i % 10] for i in range(n)]; b = [w2[i % 10] for i in range(n)]
        if n <= 200:
            ja = b"".join(jac1(xs, w) for w in a); jb = b"".join(jac2(xs, w) for w in b)
        else:
            r1 = [jac1(xs, w) for w in w1]; r2 = [jac2(xs, w) for w in w2]
            ja = b"".join(r1) * (n // 10); jb = b"".join(r2) * (n // 10)
        s1, i1 = eng.g1_sum_jac(ja, n); s2, i2 = eng.g2_sum_jac(jb, n)
        assert not i1 and not i2
        assert s1[96:] == one and s2[192:] == one + bytes(48)                  # z = FQOne / FQ2One
        assert _oracle_affine1(s1) == RC.g1_sum(b"".join(a), n) == eng.g1_sum(b"".join(a), n)
        assert _oracle_affine2(s2) == RC.g2_sum(b"".join(b), n) == eng.g2_sum(b"".join(b), n)
    # P + (-P) and the empty sum: infinity as (0, 1, 0)
    neg = w1[0][:48] + ((P.Q - int.from_bytes(w1[0][48:], "big")) % P.Q).to_bytes(48, "big")
    s, inf = eng.g1_sum_jac(jac1(xs, w1[0]) + jac1(xs, neg), 2)
    assert inf and s == g1_to_jac(None)
    s, inf = eng.g2_sum_jac(b"", 0)
    assert inf and s == g2_to_jac(None)
    # a point with z = 0 among the summands is skipped, as AddAssign does (g1.go:401-407)
    s, inf = eng.g1_sum_jac(jac1(xs, w1[0]) + g1_to_jac(w1[1], 0) + jac1(xs, w1[2]), 3)
    assert not inf and _oracle_affine1(s) == RC.g1_sum(w1[0] + w1[2], 2)


def test_prepared_keys_from_in_memory_points(eng):
    """PreparedKeys made from G2Projective records; Verify / VerifyAggregate over them with in-memory signatures"""
    xs = P.XORShift(5140)
    n = 20
    msgs, pks, sigs = _tuples("g2pubs", n, xs)
    want = [RC.g2pubs.verify(m, p, s) for m, p, s in zip(msgs, pks, sigs)]
    keys = eng.PreparedKeysJac(b"".join(jac2(xs, w) for w in pks), n)
    try:
        got = eng.g2pubs_verify_batch_prepared_jac(msgs, keys, None, b"".join(jac1(xs, w) for w in sigs))
        assert list(got) == want
        idx = np.arange(n, dtype=np.uint32)[::-1].copy()
        got = eng.g2pubs_verify_batch_prepared_jac(msgs[::-1], keys, idx, b"".join(jac1(xs, w) for w in sigs[::-1]))
        assert list(got) == want[::-1]
        sks = [sk_bytes(xs) for _ in range(n)]
        m2 = [RC.sha256(b"p%d" % i) for i in range(n)]
        k2 = eng.PreparedKeysJac(b"".join(jac2(xs, RC.g2pubs.priv_to_pub(sk)) for sk in sks), n)
        agg = RC.g1_sum(b"".join(RC.g2pubs.sign(m, sk) for m, sk in zip(m2, sks)), n)
        assert eng.g2pubs_verify_aggregate_prepared_jac(m2, k2, None, jac1(xs, agg)) is True
        assert eng.g2pubs_verify_aggregate_prepared_jac(m2[::-1], k2, None, jac1(xs, agg)) is False
        assert eng.g2pubs_verify_aggregate_prepared_jac(m2, k2, None, g1_to_jac(agg, 0)) is False
        k2.close()
    finally:
        keys.close()


def test_concurrent_single_tuple_callers_combine_by_format(eng):
    """The Go API is one tuple per call; concurrent callers are merged into one launch (verify_host.inc: Combiner).  In-memory and
    affine requests arrive interleaved from several threads: each format combines with its own kind and every caller gets its verdict."""
    xs = P.XORShift(5150)
    n = 24
    msgs, pks, sigs = _tuples("g2pubs", n, xs)
    want = [RC.g2pubs.verify(m, p, s) for m, p, s in zip(msgs, pks, sigs)]
    jp = [jac2(xs, w) for w in pks]; js = [jac1(xs, w) for w in sigs]
    got = {}
    errs = []

    def worker(i, use_jac):
        try:
            for _ in range(3):
                if use_jac:
                    ok, _ = eng.g2pubs_verify_batch_jac([msgs[i]], jp[i], js[i])
                else:
                    ok, _ = eng.g2pubs_verify_batch([msgs[i]], pks[i], sigs[i])
                got[(i, use_jac)] = bool(ok[0])
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=worker, args=(i, bool(j))) for i in range(n) for j in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs
    assert all(got[(i, j)] == want[i] for i in range(n) for j in (False, True))


def test_large_batch_of_in_memory_points_matches_the_affine_path(eng):
    """65 536 g2pubs tuples (the lane-pair kernels, BASELINE config 1's shape at config 2's size): the verdict bitmap of the in-memory
    entry point equals the affine entry point's bit for bit, and a sample equals the oracle's."""
    xs = P.XORShift(5160)
    base = 64
    msgs, pks, sigs = _tuples("g2pubs", base, xs)
    want = [RC.g2pubs.verify(m, p, s) for m, p, s in zip(msgs, pks, sigs)]
    reps = 65536 // base
    jp = b"".join(jac2(xs, w) for w in pks) * reps; js = b"".join(jac1(xs, w) for w in sigs) * reps
    allm = eng.PackedMsgs(msgs * reps)
    got, bm = eng.g2pubs_verify_batch_jac(allm, jp, js)
    aff, bm2 = eng.g2pubs_verify_batch(allm, b"".join(pks) * reps, b"".join(sigs) * reps)
    assert bytes(bm) == bytes(bm2) and list(got[:base]) == want and list(got[-base:]) == want
    assert np.array_equal(got.reshape(reps, base), np.tile(np.array(want), (reps, 1)))


def test_go_api_mirror_over_in_memory_points(eng):
    """The host mirrors (bls_amd.g2pubs / g1pubs) with their Signature / PublicKey holding points the way the reference's do
    (g2pubs/bls.go:13-15, 53-55): the flows of TestSignVerify, TestVerifyAggregate, TestVerifyAggregateCommon and the missing-signature
    test (g2pubs/bls_test.go:33-114, g1pubs/bls_test.go) -- every verdict also checked against the oracle."""
    from bls_amd import g1pubs, g2pubs
    xs = P.XORShift(5170)
    for M, R, pj, sj, newpk, newsig in ((g2pubs, RC.g2pubs, jac2, jac1, g2pubs.NewPublicKeyFromG2Projective, g2pubs.NewSignatureFromG1Projective),
                                        (g1pubs, RC.g1pubs, jac1, jac2, g1pubs.NewPublicKeyFromG1Projective, g1pubs.NewSignatureFromG2Projective)):
        n = 10
        sks = [sk_bytes(xs) for _ in range(n)]
        pkw = [R.priv_to_pub(sk) for sk in sks]
        pubs = [newpk(pj(xs, w)) for w in pkw]
        # TestSignVerify: each key signs its own message
        msgs = [b"Hello world! 16 characters %d" % i for i in range(n)]
        sgw = [R.sign(m, sk) for m, sk in zip(msgs, sks)]
        sigs = [newsig(sj(xs, w)) for w in sgw]
        assert all(M.Verify(m, p, s) for m, p, s in zip(msgs, pubs, sigs))
        assert M.VerifyBatch(msgs, pubs, sigs[1:] + sigs[:1]) == [False] * n
        assert sigs[0].s.raw == sgw[0] and pubs[3].p.raw == pkw[3]             # ToAffine().SerializeBytes() of a held point
        assert sigs[0].Serialize() == M.NewSignatureFromG1(sgw[0]).Serialize() if M is g2pubs else sigs[0].Serialize() == M.NewSignatureFromG2(sgw[0]).Serialize()
        # TestVerifyAggregate: distinct messages, AggregateSignatures over the held points (summed as Jacobian points)
        agg = M.AggregateSignatures(sigs)
        assert agg.s.jac is not None and agg.VerifyAggregate(pubs, msgs) is True
        assert R.verify_aggregate(agg.s.raw, pkw, msgs) is True
        assert agg.VerifyAggregate(pubs[1:] + pubs[:1], msgs) is False
        assert agg.VerifyAggregate(pubs, [msgs[1]] + msgs[1:]) is False        # duplicate message
        assert agg.VerifyAggregate(pubs[:-1], msgs) is False                   # length mismatch (bls.go:241-243)
        # the missing-signature test (bls_test.go:68-90): the aggregate of all but one signature does not verify
        part = M.AggregateSignatures(sigs[:-1])
        assert part.VerifyAggregate(pubs, msgs) is False and part.VerifyAggregate(pubs[:-1], msgs[:-1]) is True
        # TestVerifyAggregateCommon: one message
        msg = b"Test message common"
        csig = [newsig(sj(xs, R.sign(msg, sk))) for sk in sks]
        cagg = M.AggregateSignatures(csig)
        assert cagg.VerifyAggregateCommon(pubs, msg) is True
        assert R.verify_aggregate_common(cagg.s.raw, pkw, msg) is True
        assert cagg.VerifyAggregateCommon(pubs[:-1], msg) is False
        apk = M.AggregatePublicKeys(pubs)
        assert apk.p.jac is not None and M.Verify(msg, apk, cagg) is True
        # Copy keeps the held form; Aggregate() (bls.go:174-177) adds in place
        acc = sigs[0].Copy(); acc.Aggregate(sigs[1])
        assert acc.s.raw == (RC.g1_sum if M is g2pubs else RC.g2_sum)(sgw[0] + sgw[1], 2)
    # g1pubs WithDomain flows (g1pubs/bls_test.go)
    dom = b"\x00\x00\x00\x00\x00\x00\x00\x07"
    n = 6
    sks = [sk_bytes(xs) for _ in range(n)]
    pkw = [RC.g1pubs.priv_to_pub(sk) for sk in sks]
    pubs = [g1pubs.NewPublicKeyFromG1Projective(jac1(xs, w)) for w in pkw]
    m32 = [RC.sha256(b"d%d" % i) for i in range(n)]
    sgw = [RC.g1pubs.sign_with_domain(m, sk, dom) for m, sk in zip(m32, sks)]
    sigs = [g1pubs.NewSignatureFromG2Projective(jac2(xs, w)) for w in sgw]
    assert g1pubs.VerifyWithDomainBatch(m32, pubs, sigs, dom) == [True] * n
    assert g1pubs.VerifyWithDomain(m32[0], pubs[0], sigs[1], dom) is False
    agg = g1pubs.AggregateSignatures(sigs)
    assert g1pubs.VerifyAggregateWithDomain(agg, pubs, m32, dom) is True
    assert g1pubs.VerifyAggregateWithDomain(agg, pubs[::-1], m32, dom) is False
    csig = g1pubs.AggregateSignatures([g1pubs.NewSignatureFromG2Projective(jac2(xs, RC.g1pubs.sign_with_domain(m32[0], sk, dom))) for sk in sks])
    assert g1pubs.VerifyAggregateCommonWithDomain(csig, pubs, m32[0], dom) is True
    assert g1pubs.VerifyAggregateCommonWithDomain(csig, pubs[1:], m32[0], dom) is False
    # prepared keys from held points
    pk2 = [RC.g2pubs.priv_to_pub(sk) for sk in sks]
    keys = g2pubs.PrepareKeys([g2pubs.NewPublicKeyFromG2Projective(jac2(xs, w)) for w in pk2])
    msgs = [b"p%d" % i for i in range(n)]
    s2 = [g2pubs.NewSignatureFromG1Projective(jac1(xs, RC.g2pubs.sign(m, sk))) for m, sk in zip(msgs, sks)]
    assert g2pubs.VerifyBatchPrepared(msgs, keys, list(range(n)), s2) == [True] * n
    assert g2pubs.VerifyBatchPrepared(msgs, keys, [1] + list(range(1, n)), s2) == [False] + [True] * (n - 1)
    keys.Close()


def test_resident_in_memory_points(eng):
    """the *_jac_dev forms: Jacobian records resident in HBM -> the verdicts / Fq12 bits of the host forms"""
    import torch
    dev = torch.device("cuda", 0)
    xs = P.XORShift(5180)
    n = 300
    msgs, pks, sigs = _tuples("g2pubs", 50, xs)
    want = [RC.g2pubs.verify(m, p, s) for m, p, s in zip(msgs, pks, sigs)]
    reps = n // 50
    pm = eng.PackedMsgs(msgs * reps)
    jp = np.frombuffer(b"".join(jac2(xs, w) for w in pks) * reps, dtype=np.uint8).copy()
    js = np.frombuffer(b"".join(jac1(xs, w) for w in sigs) * reps, dtype=np.uint8).copy()
    d_m = torch.from_numpy(pm.buf.copy()).to(dev); d_o = torch.from_numpy(pm.off.view(np.int64).copy()).to(dev)
    d_p = torch.from_numpy(jp).to(dev); d_s = torch.from_numpy(js).to(dev); d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    for lat in (8192, 0):
        eng.set_latency_threshold(lat)
        d_ok.zero_()
        eng.verify_batch_jac_dev("g2pubs", d_m.data_ptr(), d_o.data_ptr(), d_p.data_ptr(), d_s.data_ptr(), d_ok.data_ptr(), n)
        assert list(d_ok.cpu().numpy().astype(bool)) == want * reps
    eng.set_latency_threshold(8192)
    w1 = [rand_g1(xs) for _ in range(4)]; w2 = [rand_g2(xs) for _ in range(4)]
    a = torch.from_numpy(np.frombuffer(b"".join(jac1(xs, w) for w in w1), dtype=np.uint8).copy()).to(dev)
    b = torch.from_numpy(np.frombuffer(b"".join(jac2(xs, w) for w in w2), dtype=np.uint8).copy()).to(dev)
    o = torch.zeros((4, 72), dtype=torch.int64, device=dev)
    eng.pairing_batch_jac_dev(a.data_ptr(), b.data_ptr(), o.data_ptr(), 4)
    assert np.array_equal(o.cpu().numpy().view(np.uint64), RC.pairing_batch(b"".join(w1), b"".join(w2), 4))


def test_sign_and_priv_to_pub_hand_back_in_memory_points(eng):
    """Sign / PrivToPub (g2pubs/bls.go:132-140, g1pubs/bls.go:132-146) with results as the Go types hold them: records with z = 1 whose
    ToAffine is the oracle's point; the zero scalar gives the reference's zero point (0, 1, 0); the records verify through the *_jac entry."""
    xs = P.XORShift(5190)
    n = 70
    sks = [sk_bytes(xs) for _ in range(n)]
    sks[3] = bytes(32)                                                         # 0 mod r: infinity
    sks[4] = P.R_ORDER.to_bytes(32, "big")
    msgs = [b"sign %d" % i for i in range(n)]
    one = mont(1).tobytes()
    for pkg, R, pk_fn, sg_fn, aff_pk, aff_sg, zero_pk, zero_sg in (
            ("g2pubs", RC.g2pubs, eng.g2_mul_generator_batch_jac, eng.g2pubs_sign_batch_jac, _oracle_affine2, _oracle_affine1, g2_to_jac(None), g1_to_jac(None)),
            ("g1pubs", RC.g1pubs, eng.g1_mul_generator_batch_jac, eng.g1pubs_sign_batch_jac, _oracle_affine1, _oracle_affine2, g1_to_jac(None), g2_to_jac(None))):
        pk = pk_fn(b"".join(sks), n); sg = sg_fn(msgs, b"".join(sks))
        for i in range(n):
            if i in (3, 4):
                assert pk[i].tobytes() == zero_pk and sg[i].tobytes() == zero_sg
                continue
            assert aff_pk(pk[i].tobytes()) == R.priv_to_pub(sks[i]) and aff_sg(sg[i].tobytes()) == R.sign(msgs[i], sks[i]), (pkg, i)
            assert one in (pk[i].tobytes()[-48:], pk[i].tobytes()[-96:-48])      # z = FQOne / FQ2One
        ver = eng.g2pubs_verify_batch_jac if pkg == "g2pubs" else eng.g1pubs_verify_batch_jac
        ok, _ = ver(msgs, pk.reshape(-1), sg.reshape(-1))
        assert list(ok) == [i not in (3, 4) for i in range(n)]
    dom = b"\x01\x00\x00\x00\x00\x00\x00\x02"
    m32 = [RC.sha256(m) for m in msgs[:5]]
    sd = eng.g1pubs_sign_with_domain_batch_jac(m32, dom, b"".join(sks[:3] + sks[5:7]))
    for i, k in enumerate(sks[:3] + sks[5:7]):
        assert _oracle_affine2(sd[i].tobytes()) == RC.g1pubs.sign_with_domain(m32[i], k, dom)
