"""-m gpu: the verify surface (BASELINE configs 1, 4, 5 at test sizes) through the host mirror of the
reference API (bls_amd.g2pubs / bls_amd.g1pubs), against the CPU oracle: hash-to-curve outputs byte for
byte, Verify verdict tables with corrupted tuples, VerifyAggregate / Common, duplicate rejection,
infinity handling, compressed wire format."""
import hashlib

import numpy as np
import pytest

from gpu_common import P, RC, rand_g1, rand_g2, sk_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["latency-path", "lane-row", "lane-quad", "lane-pair"])
def eng(request):
    """Every test of this module runs four times: small batches through the latency path (one tuple per wave, k_lat.hip), through the
    lane-quad kernels (the mid-size layout) and through the lane-pair kernels (the full-chip layout)."""
    from bls_amd import engine
    engine.init(0)
    # four paths, same results: one tuple per wave (k_lat.hip) / per lane quad (k_pairing_quad.hip) / per lane pair
    engine.set_latency_threshold(8192 if request.param in ("latency-path", "lane-row") else 0)
    engine.set_quad_threshold(0 if request.param == "lane-pair" else 16384)
    engine.set_row_threshold(*((1, 1 << 20) if request.param == "lane-row" else (0, 0)))   # round 6: sixteen lanes per tuple (k_pairing_row.hip), whatever the size
    yield engine
    engine.set_latency_threshold(8192); engine.set_quad_threshold(16384); engine.set_row_threshold(*engine.ROW_DEFAULT)


MSGS = [b"", b"a", b"the message to be signed", b"Hello world! 16 characters 0", bytes(range(55)), bytes(range(56)), bytes(range(64)), bytes(200), b"x" * 1000]


def test_hash_g1_kat_and_parity(eng, kats):
    out = eng.hash_g1_batch(MSGS)
    for i, m in enumerate(MSGS):
        assert out[i].tobytes() == RC.hash_g1(m), i
    k = kats["hash_g1"]
    assert out[2].tobytes() == int(k["x"], 16).to_bytes(48, "big") + int(k["y"], 16).to_bytes(48, "big")
    many = [b"Hello world! 16 characters %d" % i for i in range(70)]
    out = eng.hash_g1_batch(many)
    assert all(out[i].tobytes() == RC.hash_g1(m) for i, m in enumerate(many))
    # ragged batch with long messages (many SHA-256 blocks per lane, lanes finishing at different times)
    ragged = [bytes((i * 7 + j) & 0xff for j in range(n)) for i, n in enumerate((0, 63, 64, 65, 119, 120, 4096, 100003, 1))]
    out = eng.hash_g1_batch(ragged)
    assert all(out[i].tobytes() == RC.hash_g1(m) for i, m in enumerate(ragged))
    out = eng.hash_g2_batch(ragged)
    assert all(out[i].tobytes() == RC.hash_g2(m) for i, m in enumerate(ragged))


def test_hash_g2_kat_and_parity(eng, kats):
    out = eng.hash_g2_batch(MSGS)
    for i, m in enumerate(MSGS):
        assert out[i].tobytes() == RC.hash_g2(m), i
    k = kats["hash_g2"]
    exp = b"".join(int(k[f], 16).to_bytes(48, "big") for f in ("x_c0", "x_c1", "y_c0", "y_c1"))
    assert out[2].tobytes() == exp


def test_hash_g2_with_domain_kat_and_parity(eng, kats):
    msgs = [bytes(32)] + [hashlib.sha256(bytes([i])).digest() for i in range(5)]
    dom = bytes(8)
    out = eng.hash_g2_with_domain_batch(msgs, dom)
    for i, m in enumerate(msgs):
        assert out[i].tobytes() == RC.hash_g2_with_domain(m, dom), i
    assert RC.g2_compress(out[0].tobytes()).hex() == kats["hash_g2_with_domain"]["compressed_hex"]
    dom2 = bytes([1, 2, 3, 4, 5, 6, 7, 8])
    out = eng.hash_g2_with_domain_batch(msgs[:2], dom2)
    assert out[1].tobytes() == RC.hash_g2_with_domain(msgs[1], dom2)


def _tuples(group, n, seed):
    """config 1 / 5 construction: seeded keys, messages 'Hello world! 16 characters %d', every 4th tuple
    corrupted in rotation (wrong message / wrong key / negated signature)."""
    xs = P.XORShift(seed)
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    msgs, pks, sigs, expect = [], [], [], []
    for i in range(n):
        sk = sk_bytes(xs)
        m = b"Hello world! 16 characters %d" % i
        pk, sig = o.priv_to_pub(sk), o.sign(m, sk)
        good = True
        if i % 4 == 3:
            good = False
            kind = (i // 4) % 3
            if kind == 0:
                m = m + b"!"
            elif kind == 1:
                pk = o.priv_to_pub(sk_bytes(xs))
            else:
                sb = bytearray(sig)
                if group == "g2pubs":
                    sb[48:] = ((P.Q - int.from_bytes(sig[48:], "big")) % P.Q).to_bytes(48, "big")
                else:
                    for off in (96, 144):
                        sb[off:off + 48] = ((P.Q - int.from_bytes(sig[off:off + 48], "big")) % P.Q).to_bytes(48, "big")
                sig = bytes(sb)
        msgs.append(m); pks.append(pk); sigs.append(sig); expect.append(good)
    return msgs, pks, sigs, expect


@pytest.mark.parametrize("group", ["g2pubs", "g1pubs"])
def test_verify_batch_verdict_table(eng, group):
    import bls_amd.g1pubs as g1p
    import bls_amd.g2pubs as g2p
    n = 70
    msgs, pks, sigs, expect = _tuples(group, n, 1 if group == "g2pubs" else 5)
    mod = g2p if group == "g2pubs" else g1p
    mk_pk = mod.NewPublicKeyFromG2 if group == "g2pubs" else mod.NewPublicKeyFromG1
    mk_sig = mod.NewSignatureFromG1 if group == "g2pubs" else mod.NewSignatureFromG2
    got = mod.VerifyBatch(msgs, [mk_pk(p) for p in pks], [mk_sig(s) for s in sigs])
    assert got == expect
    # oracle agrees tuple by tuple (sampled: the oracle takes ~10 ms per verify)
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    for i in range(0, n, 5):
        assert o.verify(msgs[i], pks[i], sigs[i]) == expect[i]
    # single-call form and bitmap packing
    assert mod.Verify(msgs[0], mk_pk(pks[0]), mk_sig(sigs[0])) is True
    assert mod.Verify(msgs[3], mk_pk(pks[3]), mk_sig(sigs[3])) is False
    fn = eng.g2pubs_verify_batch if group == "g2pubs" else eng.g1pubs_verify_batch
    ok, bitmap = fn(msgs, b"".join(pks), b"".join(sigs))
    assert list(ok) == expect
    assert [bool(bitmap[i >> 3] >> (i & 7) & 1) for i in range(n)] == expect
    # a tuple flagged as containing the point at infinity is rejected (the reference panics there)
    flags = [0] * n; flags[0] = 1; flags[1] = 2
    ok, _ = fn(msgs, b"".join(pks), b"".join(sigs), flags)
    assert not ok[0] and not ok[1] and list(ok[2:]) == expect[2:]


@pytest.mark.parametrize("group", ["g2pubs", "g1pubs"])
def test_verify_aggregate(eng, group):
    import bls_amd.g1pubs as g1p
    import bls_amd.g2pubs as g2p
    mod = g2p if group == "g2pubs" else g1p
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    mk_pk = mod.NewPublicKeyFromG2 if group == "g2pubs" else mod.NewPublicKeyFromG1
    mk_sig = mod.NewSignatureFromG1 if group == "g2pubs" else mod.NewSignatureFromG2
    xs = P.XORShift(4)
    n = 9
    sks = [sk_bytes(xs) for _ in range(n)]
    msgs = [b">16 character identical message %d" % i for i in range(n)]
    pks = [o.priv_to_pub(sk) for sk in sks]
    sigs = [o.sign(m, sk) for m, sk in zip(msgs, sks)]
    agg = mod.AggregateSignatures([mk_sig(s) for s in sigs])
    ref_agg = (RC.g1_sum if group == "g2pubs" else RC.g2_sum)(b"".join(sigs), n)
    assert agg.s.raw == ref_agg
    pubs = [mk_pk(p) for p in pks]
    assert agg.VerifyAggregate(pubs, msgs) is True
    assert o.verify_aggregate(ref_agg, pks, msgs) is True
    assert agg.VerifyAggregate(pubs[:-1], msgs) is False                                   # length mismatch
    assert agg.VerifyAggregate(pubs, msgs[:1] + msgs[:1] + msgs[2:]) is False               # duplicate message
    assert agg.VerifyAggregate(pubs, [msgs[1], msgs[0]] + msgs[2:]) is False                # permuted messages
    assert agg.VerifyAggregate(pubs, [b""] + msgs[1:]) is False                             # empty message (nil compare upstream)
    bad = mod.AggregateSignatures([mk_sig(s) for s in sigs[:-1]])
    assert bad.VerifyAggregate(pubs, msgs) is False
    # n = 1 and n = 2 (odd/even tree shapes)
    for k in (1, 2, 3):
        a = mod.AggregateSignatures([mk_sig(s) for s in sigs[:k]])
        assert a.VerifyAggregate(pubs[:k], msgs[:k]) is True
    # common message (AggregatePublicKeys + Verify), with a missing signature (g2pubs/bls_test.go:47-90)
    msg = b">16 character identical message"
    csigs = [o.sign(msg, sk) for sk in sks]
    cagg = mod.AggregateSignatures([mk_sig(s) for s in csigs])
    assert cagg.VerifyAggregateCommon(pubs, msg) is True
    assert o.verify_aggregate_common(cagg.s.raw, pks, msg) is True
    miss = mod.AggregateSignatures([mk_sig(s) for s in csigs[1:]])
    assert miss.VerifyAggregateCommon(pubs, msg) is False
    apk = mod.AggregatePublicKeys(pubs)
    assert apk.p.raw == (RC.g2_sum if group == "g2pubs" else RC.g1_sum)(b"".join(pks), n)


def test_g1pubs_with_domain(eng):
    import bls_amd.g1pubs as g1p
    xs = P.XORShift(6)
    dom = bytes([1, 0, 0, 0, 0, 0, 0, 0])
    n = 5
    sks = [sk_bytes(xs) for _ in range(n)]
    msgs = [hashlib.sha256(bytes([i])).digest() for i in range(n)]
    pks = [RC.g1pubs.priv_to_pub(sk) for sk in sks]
    sigs = [RC.g1pubs.sign_with_domain(m, sk, dom) for m, sk in zip(msgs, sks)]
    pubs = [g1p.NewPublicKeyFromG1(p) for p in pks]; ss = [g1p.NewSignatureFromG2(s) for s in sigs]
    assert g1p.VerifyWithDomainBatch(msgs, pubs, ss, dom) == [True] * n
    assert g1p.VerifyWithDomainBatch(msgs, pubs, ss, bytes(8)) == [False] * n
    assert g1p.VerifyWithDomain(msgs[0], pubs[1], ss[0], dom) is False
    agg = g1p.AggregateSignatures(ss)
    assert g1p.VerifyAggregateWithDomain(agg, pubs, msgs, dom) is True
    assert RC.g1pubs.verify_aggregate_with_domain(agg.s.raw, pks, msgs, dom) is True
    assert g1p.VerifyAggregateWithDomain(agg, pubs, list(reversed(msgs)), dom) is False
    # no duplicate-message check in the WithDomain form (g1pubs/bls.go:300-311)
    dsigs = [RC.g1pubs.sign_with_domain(msgs[0], sk, dom) for sk in sks[:2]]
    dagg = g1p.AggregateSignatures([g1p.NewSignatureFromG2(s) for s in dsigs])
    assert g1p.VerifyAggregateWithDomain(dagg, pubs[:2], [msgs[0], msgs[0]], dom) is True
    assert g1p.VerifyAggregateCommonWithDomain(dagg, pubs[:2], msgs[0], dom) is True
    # device-side signing equals the oracle's
    mine = g1p.SignWithDomainBatch(msgs[:2], sks[:2], dom)
    assert [s.s.raw for s in mine] == sigs[:2]


def test_sign_on_device_matches_oracle(eng):
    import bls_amd.g1pubs as g1p
    import bls_amd.g2pubs as g2p
    xs = P.XORShift(1)
    sks = [sk_bytes(xs) for _ in range(3)]
    msgs = [b"Hello world! 16 characters %d" % i for i in range(3)]
    assert [s.s.raw for s in g2p.SignBatch(msgs, sks)] == [RC.g2pubs.sign(m, sk) for m, sk in zip(msgs, sks)]
    assert [s.s.raw for s in g1p.SignBatch(msgs, sks)] == [RC.g1pubs.sign(m, sk) for m, sk in zip(msgs, sks)]
    # the one-tuple forms of the Go API (g2pubs/bls.go:132-140, g1pubs/bls.go:132-146): sign, derive the key, verify
    for pkg, o in ((g2p, RC.g2pubs), (g1p, RC.g1pubs)):
        sig = pkg.Sign(msgs[0], sks[0]); pk = pkg.PrivToPub(sks[0])
        assert sig.s.raw == o.sign(msgs[0], sks[0]) and pk.p.raw == o.priv_to_pub(sks[0])
        assert pkg.Verify(msgs[0], pk, sig) is True and pkg.Verify(msgs[1], pk, sig) is False


def test_scalar_mul_and_sums(eng):
    xs = P.XORShift(3)
    n = 67
    pts1 = [rand_g1(xs) for _ in range(n)]; pts2 = [rand_g2(xs) for _ in range(n)]
    ks = [sk_bytes(xs) for _ in range(n)]
    ks[0] = bytes(32); ks[1] = (1).to_bytes(32, "big"); ks[2] = P.R_ORDER.to_bytes(32, "big"); ks[3] = (P.R_ORDER - 1).to_bytes(32, "big"); ks[4] = (2).to_bytes(32, "big")
    out, inf = eng.g1_mul_batch(b"".join(pts1), b"".join(ks), n)
    for i in range(n):
        e = RC.g1_mul(pts1[i], ks[i])
        assert (e is None) == bool(inf[i]) and (e is None or out[i].tobytes() == e), i
    out, inf = eng.g2_mul_batch(b"".join(pts2), b"".join(ks), n)
    for i in range(n):
        e = RC.g2_mul(pts2[i], ks[i])
        assert (e is None) == bool(inf[i]) and (e is None or out[i].tobytes() == e), i
    # the all-zero record is the point at infinity (the library's convention): k * infinity = infinity
    out, inf = eng.g1_mul_batch(bytes(96) + pts1[5], ks[5] + ks[5], 2)
    assert bool(inf[0]) and not bool(inf[1]) and out[0].tobytes() == bytes(96) and out[1].tobytes() == RC.g1_mul(pts1[5], ks[5])
    out, inf = eng.g2_mul_batch(bytes(192) + pts2[5], ks[5] + ks[5], 2)
    assert bool(inf[0]) and not bool(inf[1]) and out[0].tobytes() == bytes(192) and out[1].tobytes() == RC.g2_mul(pts2[5], ks[5])
    for m in (1, 2, 3, 64, 65, 67):
        assert eng.g1_sum(b"".join(pts1[:m]), m) == RC.g1_sum(b"".join(pts1[:m]), m)
        assert eng.g2_sum(b"".join(pts2[:m]), m) == RC.g2_sum(b"".join(pts2[:m]), m)
    # P + (-P) = infinity; infinity flags skip points; duplicates double
    neg = bytearray(pts1[0]); neg[48:] = ((P.Q - int.from_bytes(pts1[0][48:], "big")) % P.Q).to_bytes(48, "big")
    assert eng.g1_sum(pts1[0] + bytes(neg), 2) is None
    assert eng.g1_sum(pts1[0] + pts1[0], 2) == RC.g1_sum(pts1[0] + pts1[0], 2)
    assert eng.g1_sum(pts1[0] + pts1[1] + pts1[2], 3, [0, 1, 0]) == RC.g1_sum(pts1[0] + pts1[2], 2)
    assert eng.g2_sum(b"", 0) is None


def test_priv_to_pub_and_msm(eng):
    import bls_amd.g1pubs as g1p
    import bls_amd.g2pubs as g2p
    xs = P.XORShift(12)
    n = 66
    sks = [sk_bytes(xs) for _ in range(n)]
    sks[0] = bytes(32); sks[1] = (1).to_bytes(32, "big")
    pk2 = g2p.PrivToPubBatch(sks); pk1 = g1p.PrivToPubBatch(sks)
    assert pk2[0].p.infinity and pk1[0].p.infinity
    for i in range(1, n):
        assert pk2[i].p.raw == RC.g2pubs.priv_to_pub(sks[i]), i
        assert pk1[i].p.raw == RC.g1pubs.priv_to_pub(sks[i]), i
    # multi-scalar multiplication = sum of the per-point multiples (config 3)
    pts1 = [pk1[i].p.raw for i in range(1, n)]; pts2 = [pk2[i].p.raw for i in range(1, n)]
    ks = [sk_bytes(xs) for _ in range(n - 1)]
    ks[3] = bytes(32)
    for m in (1, 2, 5, n - 1):
        e1 = [RC.g1_mul(p, k) for p, k in zip(pts1[:m], ks[:m])]; e1 = [e for e in e1 if e is not None]
        e2 = [RC.g2_mul(p, k) for p, k in zip(pts2[:m], ks[:m])]; e2 = [e for e in e2 if e is not None]
        assert eng.g1_msm(b"".join(pts1[:m]), b"".join(ks[:m]), m) == RC.g1_sum(b"".join(e1), len(e1))
        assert eng.g2_msm(b"".join(pts2[:m]), b"".join(ks[:m]), m) == RC.g2_sum(b"".join(e2), len(e2))
    assert eng.g1_msm(b"", b"", 0) is None
    assert eng.g1_msm(pts1[0], bytes(32), 1) is None


def test_concurrent_callers(eng):
    """The C ABI is callable from several OS threads at once (cgo pins one per call): each call leases its own
    stream/scratch context.  ctypes drops the GIL for the duration of a call, so these really overlap."""
    import threading
    import bls_amd.g1pubs as g1p
    import bls_amd.g2pubs as g2p
    jobs = []
    for group, seed in (("g2pubs", 21), ("g1pubs", 22), ("g2pubs", 23), ("g1pubs", 24)):
        msgs, pks, sigs, expect = _tuples(group, 24, seed)
        jobs.append((group, msgs, pks, sigs, expect))
    xs = P.XORShift(25)
    g1s = [rand_g1(xs) for _ in range(8)]; g2s = [rand_g2(xs) for _ in range(8)]
    want_pair = eng.pairing_batch(b"".join(g1s), b"".join(g2s), 8).copy()
    results = {}

    def run(idx):
        try:
            if idx < len(jobs):
                group, msgs, pks, sigs, expect = jobs[idx]
                mod = g2p if group == "g2pubs" else g1p
                mk_pk = mod.NewPublicKeyFromG2 if group == "g2pubs" else mod.NewPublicKeyFromG1
                mk_sig = mod.NewSignatureFromG1 if group == "g2pubs" else mod.NewSignatureFromG2
                for _ in range(3):
                    got = mod.VerifyBatch(msgs, [mk_pk(p) for p in pks], [mk_sig(s) for s in sigs])
                    assert got == expect
            else:
                for _ in range(6):
                    assert (eng.pairing_batch(b"".join(g1s), b"".join(g2s), 8) == want_pair).all()
            results[idx] = True
        except Exception as e:  # noqa: BLE001
            results[idx] = e

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs) + 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(results.get(i) is True for i in range(len(threads))), results


def test_concurrent_with_domain_callers_two_domains(eng):
    """Concurrent small g1pubs VerifyWithDomain calls are combined per DOMAIN (g1pubs/bls.go:171-174): sixteen threads, two
    domains, valid and corrupted tuples interleaved; every caller gets its own verdicts."""
    import threading
    xs = P.XORShift(77)
    doms = [bytes([1, 0, 0, 0, 0, 0, 0, 0]), bytes([2, 9, 0, 0, 0, 0, 0, 7])]
    jobs = []
    for i in range(16):
        dom = doms[i % 2]
        sk = sk_bytes(xs)
        m32 = bytes([i]) * 32
        pk = RC.g1pubs.priv_to_pub(sk)
        sig = RC.g1pubs.sign_with_domain(m32, sk, dom)
        if i % 5 == 4:
            m_use, want = bytes([i + 1]) * 32, False                      # wrong message
        elif i % 7 == 6:
            m_use, want = m32, None                                        # verified under the OTHER domain below -> False
        else:
            m_use, want = m32, True
        jobs.append((m_use, dom if want is not None else doms[(i + 1) % 2], pk, sig, bool(want)))
    results = {}

    def run(idx):
        try:
            m, dom, pk, sig, want = jobs[idx]
            for _ in range(4):
                got = eng.g1pubs_verify_with_domain_batch([m], dom, pk, sig)
                assert bool(got[0]) is want, (idx, got)
            results[idx] = True
        except Exception as e:  # noqa: BLE001
            results[idx] = e
    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(results.get(i) is True for i in range(len(jobs))), results
    assert RC.g1pubs.verify_with_domain(jobs[0][0], jobs[0][2], jobs[0][3], jobs[0][1]) is True


def test_wire_format(eng, kats):
    xs = P.XORShift(9)
    n = 10
    p1 = [rand_g1(xs) for _ in range(n)]; p2 = [rand_g2(xs) for _ in range(n)]
    c1 = eng.g1_compress_batch(b"".join(p1), n); c2 = eng.g2_compress_batch(b"".join(p2), n)
    assert [c.tobytes() for c in c1] == [RC.g1_compress(p) for p in p1]
    assert [c.tobytes() for c in c2] == [RC.g2_compress(p) for p in p2]
    out, inf, err = eng.g1_decompress_batch(c1.reshape(-1), n)
    assert not err.any() and not inf.any() and [o.tobytes() for o in out] == p1
    out, inf, err = eng.g2_decompress_batch(c2.reshape(-1), n)
    assert not err.any() and not inf.any() and [o.tobytes() for o in out] == p2
    # error table: the reference's rejected encodings (g2pubs/bls_test.go:336-347, g1pubs/bls_test.go:422-433) + crafted
    bad2 = bytes.fromhex(kats["invalid_pubkey_g2pubs_hex"]); bad1 = bytes.fromhex(kats["invalid_pubkey_g1pubs_hex"])
    inf1 = RC.g1_compress(None); junk = bytearray(inf1); junk[7] = 1
    cases1 = [bad1, bytes(48), inf1, bytes(junk), c1[0].tobytes()]
    out, inf, err = eng.g1_decompress_batch(b"".join(cases1), len(cases1))
    exp = [RC.g1_decompress(c) for c in cases1]
    assert [int(e) for e in err] == [e for e, _ in exp]
    assert bool(inf[2]) and err[2] == 0
    out2, inf2, err2 = eng.g2_decompress_batch(bad2 + RC.g2_compress(None), 2)
    assert int(err2[0]) == RC.g2_decompress(bad2)[0] != 0 and bool(inf2[1]) and err2[1] == 0
    # a curve point outside the prime-order subgroup is rejected only when the check is on
    x = 0
    while True:
        pt = P.g1_from_x(x, False)
        if pt is not None and not P.g1_in_subgroup(pt):
            break
        x += 1
    comp = P.g1_compress(pt)
    _, _, e_on = eng.g1_decompress_batch(comp, 1, True)
    _, _, e_off = eng.g1_decompress_batch(comp, 1, False)
    assert int(e_on[0]) == 4 and int(e_off[0]) == 0
    # the endomorphism form of the subgroup test (check = 1) and the reference's r * P form (check = 2) agree on curve
    # points outside the subgroup (small x) and inside it, for both groups; the oracle agrees on a sample
    outs1, outs2 = [], []
    for x in range(60):
        pt = P.g1_from_x(x, bool(x & 1))
        if pt is not None:
            outs1.append((P.g1_compress(pt), P.g1_in_subgroup(pt)))
    for x0 in range(8):
        for x1 in range(8):
            pt = P.g2_from_x((x0, x1), bool(x0 & 1))
            if pt is not None:
                outs2.append((P.g2_compress(pt), None))
    assert len(outs1) > 15 and len(outs2) > 15
    for dec, cases, extra in ((eng.g1_decompress_batch, outs1, c1), (eng.g2_decompress_batch, outs2, c2)):
        blob = b"".join(c for c, _ in cases) + extra.tobytes()
        m = len(cases) + n
        _, _, e1 = dec(blob, m, 1)
        _, _, e2 = dec(blob, m, 2)
        assert [int(v) for v in e1] == [int(v) for v in e2]
        assert all(int(v) == 0 for v in e1[len(cases):])                   # genuine subgroup points pass
        assert sum(int(v) == 4 for v in e1[:len(cases)]) >= len(cases) - 2  # small-x points are (almost) never in the subgroup
        for k, (c, insub) in enumerate(cases):
            if insub is not None:
                assert (int(e1[k]) == 0) == insub
    assert P.g2_in_subgroup(P.g2_from_x((0, 1), False)) == (int(eng.g2_decompress_batch(P.g2_compress(P.g2_from_x((0, 1), False)), 1, 1)[2][0]) == 0) if P.g2_from_x((0, 1), False) is not None else True
    from bls_amd import g2pubs
    with pytest.raises(g2pubs.DeserializeError):
        g2pubs.DeserializePublicKey(bad2)
    pk = g2pubs.DeserializePublicKey(c2[0].tobytes())
    assert pk.p.raw == p2[0] and pk.Serialize() == c2[0].tobytes()


def test_sharded_verify_aggregate_single_process(eng):
    """The multi-GPU VerifyAggregate protocol (bls_amd/dist.py) with both shards evaluated on this GPU: the
    all-gathers are simulated by computing every rank's contribution first."""
    from bls_amd import dist as bdist
    xs = P.XORShift(4)
    n = 11
    sks = [sk_bytes(xs) for _ in range(n)]
    msgs = [b">16 character identical message %d" % i for i in range(n)]
    for group, o, sumf in [("g2pubs", RC.g2pubs, RC.g1_sum), ("g1pubs", RC.g1pubs, RC.g2_sum)]:
        pks = [o.priv_to_pub(sk) for sk in sks]
        agg = sumf(b"".join(o.sign(m, sk) for m, sk in zip(msgs, sks)), n)
        for world in (2, 3):
            def run(pk_list):
                logs = {}

                def gather_factory(rank):
                    calls = {"i": 0}

                    def gather(b):
                        i = calls["i"]; calls["i"] += 1
                        logs.setdefault(i, {})[rank] = b
                        return None
                    return gather
                # pass 1: record each rank's contributions; pass 2: replay with full gathers
                contrib = []
                fkey = (np.uint64(0x1234567890abcdef), np.uint64(0xfedcba0987654321))   # the nonce the ranks agree on: XOR of their contributions
                knonce = [b"K" + np.array(fkey, dtype=np.uint64).tobytes()] + [b"K" + bytes(16)] * (world - 1)
                for r in range(world):
                    lo, hi = bdist.shard_bounds(n, r, world)
                    keys = np.frombuffer(bdist.message_keys(msgs[lo:hi]), dtype=np.uint8).reshape(-1, 33)
                    dig = bdist.row_fingerprints(keys, fkey).tobytes() + b"\x00"    # fingerprints + status byte
                    part = eng.aggregate_partial(group, msgs[lo:hi], b"".join(pk_list[lo:hi]))[0].tobytes() + b"\x00"
                    contrib.append((dig, part))
                outs = []
                for r in range(world):
                    lo, hi = bdist.shard_bounds(n, r, world)
                    # the four (tagged) exchanges of a duplicate-free aggregate: nonce, fingerprints, suspicion flags, partial products
                    seq = iter([knonce, [b"F" + c[0] for c in contrib], [b"S\x00"] * world, [b"P" + c[1] for c in contrib]])
                    outs.append(bdist.sharded_verify_aggregate(group, msgs[lo:hi], b"".join(pk_list[lo:hi]), agg, r, world, lambda b: next(seq)))
                assert len(set(outs)) == 1
                return outs[0]
            assert run(pks) is True
            assert run([pks[1], pks[0]] + pks[2:]) is False
    # partial product of an empty shard is 1, and fq12_product agrees with the oracle
    one, bad = eng.aggregate_partial("g2pubs", [], b"")
    assert not bad
    assert np.array_equal(eng.fq12_product(np.stack([one, one])), one)


def test_rccl_collectives_world1(eng):
    """The production collectives (torch.distributed backend "nccl" = RCCL, device tensors) on a one-rank group:
    bitmap all-reduce of a sharded batch verify and the two all-gathers of a sharded VerifyAggregate."""
    import os
    import torch
    import torch.distributed as dist
    from bls_amd import dist as bdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        msgs, pks, sigs, expect = _tuples("g2pubs", 19, 31)

        def shard(lo, hi):
            ok, _ = eng.g2pubs_verify_batch(msgs[lo:hi], b"".join(pks[lo:hi]), b"".join(sigs[lo:hi]))
            return ok
        bitmap = bdist.sharded_verify_bitmap(len(msgs), shard, 0, 1, bdist.torch_all_reduce(dev))
        assert list(bdist.unpack_bitmap(bitmap, len(msgs))) == expect
        xs = P.XORShift(32)
        sks = [sk_bytes(xs) for _ in range(5)]
        amsgs = [b"distinct message %d" % i for i in range(5)]
        apks = [RC.g2pubs.priv_to_pub(sk) for sk in sks]
        agg = RC.g1_sum(b"".join(RC.g2pubs.sign(m, sk) for m, sk in zip(amsgs, sks)), 5)
        gather = bdist.torch_all_gather_bytes(dev)
        assert bdist.sharded_verify_aggregate("g2pubs", amsgs, b"".join(apks), agg, 0, 1, gather) is True
        assert bdist.sharded_verify_aggregate("g2pubs", amsgs, b"".join(apks[::-1]), agg, 0, 1, gather) is False
    finally:
        dist.destroy_process_group()


def test_concurrent_single_tuple_calls_are_combined(eng):
    """The Go API verifies one tuple per call; concurrent callers are merged into shared batch launches by the
    library (verify_host.inc, request combining).  Every caller must still get exactly its own verdicts."""
    import threading
    msgs, pks, sigs, expect = _tuples("g2pubs", 48, 41)
    m1, p1, s1, e1 = _tuples("g1pubs", 16, 42)
    results = {}

    def run(idx):
        try:
            out = []
            if idx < 48:
                for rep in range(3):
                    ok, bitmap = eng.g2pubs_verify_batch([msgs[idx]], pks[idx], sigs[idx])
                    out.append(bool(ok[0]) == expect[idx] and bool(bitmap[0] & 1) == expect[idx])
                j = (idx * 7) % 46                                                   # a 3-tuple request in the same queue
                ok, _ = eng.g2pubs_verify_batch(msgs[j:j + 3], b"".join(pks[j:j + 3]), b"".join(sigs[j:j + 3]))
                out.append(list(ok) == expect[j:j + 3])
            else:
                k = idx - 48
                for rep in range(3):
                    ok, _ = eng.g1pubs_verify_batch([m1[k]], p1[k], s1[k])
                    out.append(bool(ok[0]) == e1[k])
            results[idx] = all(out)
        except Exception as e:  # noqa: BLE001
            results[idx] = e

    threads = [threading.Thread(target=run, args=(i,)) for i in range(64)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(results.get(i) is True for i in range(64)), {i: r for i, r in results.items() if r is not True}


def test_lifecycle_and_argument_errors(eng):
    """blsmi_shutdown / re-init keeps working (contexts, generator tables and pools are rebuilt), bad arguments are
    reported as error codes, never crashes."""
    import ctypes as C
    from bls_amd import _native
    lib = _native.load()
    msgs, pks, sigs, expect = _tuples("g2pubs", 5, 51)
    ok, _ = eng.g2pubs_verify_batch(msgs, b"".join(pks), b"".join(sigs))
    assert list(ok) == expect
    lib.blsmi_shutdown()
    lib.blsmi_shutdown()                                                   # idempotent
    ok, _ = eng.g2pubs_verify_batch(msgs, b"".join(pks), b"".join(sigs))   # lazily re-initialises
    assert list(ok) == expect
    assert lib.blsmi_init(0) == 0
    assert lib.blsmi_init(99) != 0                                         # another device than the one in use
    out = (C.c_uint64 * 72)()
    assert lib.blsmi_pairing_batch(None, None, out, C.c_size_t(1)) != 0    # NULL inputs with n > 0
    assert lib.blsmi_pairing_batch(None, None, None, C.c_size_t(0)) == 0   # empty batch is fine
    okb = (C.c_uint8 * 1)()
    assert lib.blsmi_g2pubs_verify_batch(None, None, None, None, None, okb, None, C.c_size_t(1)) != 0
    assert lib.blsmi_last_kernel_ms(None, None) != 0


@pytest.mark.parametrize("group", ["g2pubs", "g1pubs"])
def test_verify_serialized_batch(eng, group, kats):
    """Deserialize + Verify in one pass over the compressed wire format equals deserialising with the oracle and
    verifying; malformed / off-curve / infinity encodings give verdict False with the right error code."""
    import bls_amd.g1pubs as g1p
    import bls_amd.g2pubs as g2p
    mod = g2p if group == "g2pubs" else g1p
    msgs, pks, sigs, expect = _tuples(group, 37, 61)
    cpk = RC.g2_compress if group == "g2pubs" else RC.g1_compress
    csg = RC.g1_compress if group == "g2pubs" else RC.g2_compress
    pkc = [cpk(p) for p in pks]; sgc = [csg(s) for s in sigs]
    assert mod.VerifySerializedBatch(msgs, pkc, sgc) == expect
    # damage some encodings: cleared compression bit, infinity, x not on the curve (the reference's rejected vectors)
    bad_pk = bytes.fromhex(kats["invalid_pubkey_g2pubs_hex" if group == "g2pubs" else "invalid_pubkey_g1pubs_hex"])
    pkc2 = list(pkc); sgc2 = list(sgc); exp2 = list(expect)
    pkc2[0] = bad_pk; exp2[0] = False
    b = bytearray(sgc2[1]); b[0] &= 0x7f; sgc2[1] = bytes(b); exp2[1] = False
    pkc2[2] = cpk(None); exp2[2] = False
    sgc2[4] = csg(None); exp2[4] = False
    ok, ep, es = eng.verify_serialized_batch(group, msgs, b"".join(pkc2), b"".join(sgc2))
    assert list(ok) == exp2
    dpk = RC.g2_decompress if group == "g2pubs" else RC.g1_decompress
    dsg = RC.g1_decompress if group == "g2pubs" else RC.g2_decompress
    assert [int(e) for e in ep] == [dpk(c)[0] for c in pkc2]
    assert [int(e) for e in es] == [dsg(c)[0] for c in sgc2]
    assert ep[0] != 0 and es[1] != 0 and ep[2] == 0 and es[4] == 0      # infinity deserialises, but cannot be verified


def test_msm_bucket_method(eng):
    """From 2^17 points on, blsmi_g{1,2}_msm runs the bucket method (msm.inc).  Same group element as the per-point
    multiples summed up -- checked against that device path (itself oracle-checked above) and against the oracle's
    closed form sum k_i (b_i G) = (sum k_i b_i) G, with repeated points, zero and maximal scalars in the mix."""
    rng = np.random.default_rng(77)
    for n in (1 << 17, (1 << 17) + 5003):
        base = 257
        bk = rng.integers(0, 256, size=(base, 32), dtype=np.uint8); bk[:, 0] &= 0x3f
        k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        k[0] = 0; k[1] = 255; k[2] = 0; k[2, 31] = 1; k[3] = k[4]                    # zero, 2^256-1, one, a repeated scalar
        for gen, mul, summ, msm, ref_mul in ((RC.g1_generator(), eng.g1_mul_batch, eng.g1_sum, eng.g1_msm, RC.g1_mul),
                                              (RC.g2_generator(), eng.g2_mul_batch, eng.g2_sum, eng.g2_msm, RC.g2_mul)):
            bpts, _ = mul(gen * base, bk.reshape(-1), base)
            pts = bpts[np.arange(n) % base]                                           # every base point ~20 times
            got = msm(pts.reshape(-1), k.reshape(-1), n)
            mults, inf = mul(pts.reshape(-1), k.reshape(-1), n)
            assert got == summ(mults.reshape(-1), n, inf.astype(np.uint8))
            kw = k.reshape(n, 8, 4).astype(np.uint64)                                   # sum of k_i per base point, 32-bit words
            words = (kw[:, :, 0] << 24) | (kw[:, :, 1] << 16) | (kw[:, :, 2] << 8) | kw[:, :, 3]
            acc = 0
            for j in range(base):
                sel = words[j::base]
                tot = 0
                for w in range(8):
                    tot = (tot << 32) + int(sel[:, w].sum())
                acc += tot * int.from_bytes(bk[j].tobytes(), "big")
            assert got == ref_mul(gen, (acc % P.R_ORDER).to_bytes(32, "big"))
            if n != 1 << 17:
                # degenerate digits (every scalar equal): the bucket pass would be one lane per window; the library
                # detects the skew and takes the scalar-independent path -- same answer: k * sum(P_i)
                same = np.tile(k[5], (n, 1))
                tot = summ(pts.reshape(-1), n)
                assert msm(pts.reshape(-1), same.reshape(-1), n) == ref_mul(tot, k[5].tobytes())
