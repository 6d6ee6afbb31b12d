"""Cross-check the C oracle against the independent Python big-int twin on seeded random inputs,
and replay the reference's deterministic self-consistency tests (g2pubs/bls_test.go:33-185,
g1pubs/bls_test.go) through the C oracle."""
import numpy as np
import pytest

from oracle import pyref as P
from oracle import refcpu as RC


def mont(v):
    return np.array(P.limbs64(P.to_mont(v)), dtype=np.uint64)


def unmont(l):
    return P.from_mont(P.from_limbs64(l))


def rfq(xs):
    return P.rand_int(xs, P.Q)


def pack(vals):
    return np.concatenate([mont(v) for v in vals])


def unpack(arr):
    return [unmont(arr[6 * i:6 * i + 6]) for i in range(len(arr) // 6)]


def flat6(a):
    return [a[j][k] for j in range(3) for k in range(2)]


def to6(v):
    return tuple((v[2 * j], v[2 * j + 1]) for j in range(3))


def to12(v):
    return (to6(v[:6]), to6(v[6:]))


def test_fq_ops_vs_python():
    xs = P.XORShift(1)
    for _ in range(200):
        a, b = rfq(xs), rfq(xs)
        assert unmont(RC.fq_mul(mont(a), mont(b))) == a * b % P.Q
        assert unmont(RC.fq_sqr(mont(a))) == a * a % P.Q
        assert unmont(RC.fq_add(mont(a), mont(b))) == (a + b) % P.Q
        assert unmont(RC.fq_sub(mont(a), mont(b))) == (a - b) % P.Q
        assert unmont(RC.fq_neg(mont(a))) == (-a) % P.Q
        assert unmont(RC.fq_dbl(mont(a))) == 2 * a % P.Q
        ok, inv = RC.fq_inverse(mont(a)); assert ok and unmont(inv) == pow(a, -1, P.Q)
        ok, s = RC.fq_sqrt(mont(a * a % P.Q)); assert ok and unmont(s) == P.fq_sqrt(a * a % P.Q)
        assert P.from_limbs64(RC.fq_to_repr(mont(a))) == a
        assert list(RC.fq_from_repr(P.limbs64(a))) == list(mont(a))
    assert RC.fq_inverse(np.zeros(6, np.uint64))[0] == 0
    # FQReprToFQ of an out-of-range repr is zero (fq.go:49-56)
    assert not RC.fq_from_repr(P.limbs64(P.Q)).any()


def test_tower_ops_vs_python():
    xs = P.XORShift(3)
    for _ in range(20):
        a6 = [rfq(xs) for _ in range(6)]; b6 = [rfq(xs) for _ in range(6)]
        assert unpack(RC.fq6_mul(pack(a6), pack(b6))) == flat6(P.fq6_mul(to6(a6), to6(b6)))
        assert unpack(RC.fq6_sqr(pack(a6))) == flat6(P.fq6_sqr(to6(a6)))
        ok, inv = RC.fq6_inverse(pack(a6)); assert ok and unpack(inv) == flat6(P.fq6_inv(to6(a6)))
        a12 = [rfq(xs) for _ in range(12)]; b12 = [rfq(xs) for _ in range(12)]
        assert unpack(RC.fq12_mul(pack(a12), pack(b12))) == P.fq12_flat(P.fq12_mul(to12(a12), to12(b12)))
        assert unpack(RC.fq12_sqr(pack(a12))) == P.fq12_flat(P.fq12_sqr(to12(a12)))
        ok, inv = RC.fq12_inverse(pack(a12)); assert ok and unpack(inv) == P.fq12_flat(P.fq12_inv(to12(a12)))
        c = [(rfq(xs), rfq(xs)) for _ in range(3)]
        got = RC.fq12_mul_by_014(pack(a12), pack(c[0]), pack(c[1]), pack(c[2]))
        assert unpack(got) == P.fq12_flat(P.fq12_mul_by_014(to12(a12), *c))
        # sparse == dense (fq12_test.go:10-39)
        dense = ((c[0], c[1], (0, 0)), ((0, 0), c[2], (0, 0)))
        assert unpack(got) == P.fq12_flat(P.fq12_mul(to12(a12), dense))
        for p in range(0, 13):
            assert unpack(RC.fq12_frobenius(pack(a12), p)) == P.fq12_flat(P.fq12_frob(to12(a12), p))
            assert unpack(RC.fq6_frobenius(pack(a6), p)) == flat6(P.fq6_frob(to6(a6), p))
    # frobenius == x^q (fq12_test.go:54-72)
    a12 = [rfq(xs) for _ in range(12)]
    assert P.fq12_frob(to12(a12), 1) == P.fq12_pow(to12(a12), P.Q)


def test_curve_and_pairing_vs_python():
    xs = P.XORShift(2)
    g1b, g2b = RC.g1_generator(), RC.g2_generator()
    for _ in range(4):
        a, b = P.rand_fr(xs), P.rand_fr(xs)
        pa = P.jac_to_affine(P.F1, P.affine_mul(P.F1, P.G1_GEN, a))
        qb = P.jac_to_affine(P.F2, P.affine_mul(P.F2, P.G2_GEN, b))
        assert RC.g1_mul(g1b, a.to_bytes(32, "big")) == P.g1_serialize(pa)
        assert RC.g2_mul(g2b, b.to_bytes(32, "big")) == P.g2_serialize(qb)
        # prepared coefficients and Miller-loop output, bit for bit
        prep = RC.g2_prepare(P.g2_serialize(qb))
        pyprep = P.g2_prepare(qb)
        assert len(pyprep) == 68
        for i in range(68):
            for j in range(3):
                assert (unmont(prep[i, j, :6]), unmont(prep[i, j, 6:])) == pyprep[i][j]
        ml = RC.miller_loop(P.g1_serialize(pa), P.g2_serialize(qb), 1)
        assert unpack(ml) == P.fq12_flat(P.miller_loop([(pa, pyprep)]))
        e = RC.pairing_batch(P.g1_serialize(pa), P.g2_serialize(qb), 1)[0]
        assert unpack(e) == P.fq12_flat(P.pairing(pa, qb))
        ok, fe = RC.final_exponentiation(ml); assert ok and list(fe) == list(e)
    # bilinearity ties random pairings to the single reference KAT
    kat = P.pairing(P.G1_GEN, P.G2_GEN)
    assert P.pairing(pa, qb) == P.fq12_pow(kat, a * b % P.R_ORDER)
    # compress / decompress round trips and subgroup checks
    assert RC.g1_decompress(RC.g1_compress(P.g1_serialize(pa))) == (0, P.g1_serialize(pa))
    assert RC.g2_decompress(RC.g2_compress(P.g2_serialize(qb))) == (0, P.g2_serialize(qb))
    assert RC.g1_compress(P.g1_serialize(pa)) == P.g1_compress(pa) and RC.g2_compress(P.g2_serialize(qb)) == P.g2_compress(qb)
    assert RC.g1_decompress(RC.g1_compress(None)) == (0, None) and RC.g2_decompress(RC.g2_compress(None)) == (0, None)
    assert RC.g1_decompress(bytes(48))[0] == 1            # compression bit clear
    bad = bytearray(RC.g1_compress(None)); bad[5] = 1
    assert RC.g1_decompress(bytes(bad))[0] == 2           # junk in infinity encoding


def test_hash_vs_python():
    for i in range(6):
        m = b"Hello world! 16 characters %d" % i
        assert RC.hash_g1(m) == P.g1_serialize(P.hash_g1(m))
        assert RC.hash_g2(m) == P.g2_serialize(P.hash_g2(m))
    for m in [b"", b"a" * 200]:
        assert RC.hash_g1(m) == P.g1_serialize(P.hash_g1(m))
    d = bytes(range(8))
    m32 = bytes(range(32))
    assert RC.hash_g2_with_domain(m32, d) == P.g2_serialize(P.jac_to_affine(P.F2, P.hash_g2_with_domain(m32, d)))


# ---- replay of the reference's API-level tests through the C oracle ------------------------------
def sk_bytes(xs):
    return P.rand_fr(xs).to_bytes(32, "big")


def test_g2pubs_sign_verify_replay():
    # g2pubs/bls_test.go:33-45 (seed 1)
    xs = P.XORShift(1)
    for i in range(3):
        sk = sk_bytes(xs)
        pk = RC.g2pubs.priv_to_pub(sk)
        msg = b"Hello world! 16 characters %d" % i
        sig = RC.g2pubs.sign(msg, sk)
        assert RC.g2pubs.verify(msg, pk, sig)
        assert not RC.g2pubs.verify(msg + b"!", pk, sig)
        if i == 0:     # python twin agrees on the signature bytes and the verdicts
            skv = int.from_bytes(sk, "big")
            assert P.g2_serialize(P.G2Pubs.priv_to_pub(skv)) == pk and P.g1_serialize(P.G2Pubs.sign(msg, skv)) == sig


def test_g2pubs_aggregate_replay():
    # common message, with a missing signature (g2pubs/bls_test.go:47-90; seeds 2, 3)
    xs = P.XORShift(3)
    msg = b">16 character identical message"
    n, skipped = 4, 2
    pks, sigs = [], []
    for i in range(n):
        sk = sk_bytes(xs)
        pks.append(RC.g2pubs.priv_to_pub(sk))
        if i != skipped:
            sigs.append(RC.g2pubs.sign(msg, sk))
        agg = RC.g1_sum(b"".join(sigs), len(sigs)) if sigs else None
        if agg is not None:
            assert RC.g2pubs.verify_aggregate_common(agg, pks, msg) == (i < skipped)
    # distinct messages (bls_test.go:92-114; seed 4) + duplicate rejection (:144-185; seed 5)
    xs = P.XORShift(4)
    pks, sigs, msgs = [], [], []
    for i in range(3):
        sk = sk_bytes(xs)
        m = b">16 character identical message %d" % i
        pks.append(RC.g2pubs.priv_to_pub(sk)); msgs.append(m); sigs.append(RC.g2pubs.sign(m, sk))
    agg = RC.g1_sum(b"".join(sigs), 3)
    assert RC.g2pubs.verify_aggregate(agg, pks, msgs)
    assert not RC.g2pubs.verify_aggregate(agg, pks[:2], msgs)                 # length mismatch
    assert not RC.g2pubs.verify_aggregate(agg, pks, [msgs[0], msgs[0], msgs[2]])  # duplicate message
    assert not RC.g2pubs.verify_aggregate(agg, pks, [msgs[1], msgs[0], msgs[2]])  # wrong pairing of msgs
    # python twin: same verdicts on the same inputs
    def g1p(b): return (int.from_bytes(b[:48], "big"), int.from_bytes(b[48:], "big"))
    def g2p(b): return ((int.from_bytes(b[:48], "big"), int.from_bytes(b[48:96], "big")), (int.from_bytes(b[96:144], "big"), int.from_bytes(b[144:], "big")))
    assert P.G2Pubs.verify_aggregate(g1p(agg), [g2p(p) for p in pks], msgs)
    # an empty message is rejected by the duplicate check's nil comparison (g2pubs/bls.go:249-258)
    assert not RC.g2pubs.verify_aggregate(agg, pks, [b"", msgs[1], msgs[2]])


def test_g1pubs_replay():
    xs = P.XORShift(1)
    sk = sk_bytes(xs)
    pk = RC.g1pubs.priv_to_pub(sk)
    msg = b"Hello world! 16 characters 0"
    sig = RC.g1pubs.sign(msg, sk)
    assert RC.g1pubs.verify(msg, pk, sig) and not RC.g1pubs.verify(b"x" + msg, pk, sig)
    skv = int.from_bytes(sk, "big")
    assert P.g1_serialize(P.G1Pubs.priv_to_pub(skv)) == pk and P.g2_serialize(P.G1Pubs.sign(msg, skv)) == sig
    # WithDomain (g1pubs/bls_test.go, verify_benchmark_test.go)
    m32, dom = bytes(range(32)), bytes([1, 0, 0, 0, 0, 0, 0, 0])
    sigd = RC.g1pubs.sign_with_domain(m32, sk, dom)
    assert RC.g1pubs.verify_with_domain(m32, pk, sigd, dom)
    assert not RC.g1pubs.verify_with_domain(m32, pk, sigd, bytes(8))
    sk2 = sk_bytes(xs); pk2 = RC.g1pubs.priv_to_pub(sk2)
    m32b = bytes(range(1, 33))
    agg = RC.g2_sum(sigd + RC.g1pubs.sign_with_domain(m32b, sk2, dom), 2)
    assert RC.g1pubs.verify_aggregate_with_domain(agg, [pk, pk2], [m32, m32b], dom)
    assert not RC.g1pubs.verify_aggregate_with_domain(agg, [pk2, pk], [m32, m32b], dom)
    aggc = RC.g2_sum(sigd + RC.g1pubs.sign_with_domain(m32, sk2, dom), 2)
    assert RC.g1pubs.verify_aggregate_common_with_domain(aggc, [pk, pk2], m32, dom)
    sig2 = RC.g1pubs.sign(b"other", sk2)
    agg2 = RC.g2_sum(sig + sig2, 2)
    assert RC.g1pubs.verify_aggregate(agg2, [pk, pk2], [msg, b"other"])
    assert not RC.g1pubs.verify_aggregate(agg2, [pk, pk2], [msg, msg])
