"""-m gpu: parity cases added in round 3 (VERDICT r02 "next round" items 4 and 5).

* the device-pointer forms of scalar multiplication, point sums, MSM and VerifyAggregate (blsmi 0.3) against the host
  forms and the oracle -- including the duplicate-message rejection that runs on the device for resident messages;
* non-canonical field encodings (x >= q) through the decompressions, Deserialize+Verify and the affine C-ABI inputs:
  FQReprToFQ maps such a value to 0 (fq.go:49-56), so DecompressG1 of it yields (0, +-2), a curve point outside the
  subgroup (g1.go:199-227);
* points ON the curve but OUTSIDE the prime-order subgroup through pairing_batch / verify_batch on both paths (the latency
  programs and the throughput Miller loops use homogeneous steps whose Miller value differs from g2.go:655-772 by
  subfield factors; the final exponentiation must remove them for every curve point, not only for subgroup points);
* VerifyAggregateWithDomain at n >= 8192 (g1pubs/bls.go:300-311; reaches the two-tuples-per-loop kernel);
* a seeded slice of tools/soak.py (differential run of the latency programs against the throughput kernels).
"""
import hashlib
import os
import subprocess
import sys
import json

import numpy as np
import pytest

from gpu_common import P, RC, rand_g1, rand_g2, sk_bytes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = P.Q


@pytest.fixture(scope="module", params=["latency-path", "lane-quad", "lane-pair"])
def eng(request):
    from bls_amd import engine
    engine.init(0)
    # three paths, same results: one tuple per wave (k_lat.hip) / per lane quad (k_pairing_quad.hip) / per lane pair
    engine.set_latency_threshold(8192 if request.param == "latency-path" else 0)
    engine.set_quad_threshold(0 if request.param == "lane-pair" else 16384)
    engine.set_row_threshold(0, 0)                                         # (the lane-row layout has its own module, tests/test_gpu_row.py, and the fixtures of test_gpu_pairing / verify / prepared / jac)
    yield engine
    engine.set_latency_threshold(8192); engine.set_quad_threshold(16384); engine.set_row_threshold(*engine.ROW_DEFAULT)


def _dev(x):
    import torch
    a = np.frombuffer(x, dtype=np.uint8).copy() if isinstance(x, (bytes, bytearray)) else np.array(x)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    return torch.from_numpy(a).to(torch.device("cuda", 0))


def _neg_g1(p):
    return p[:48] + ((Q - int.from_bytes(p[48:], "big")) % Q).to_bytes(48, "big")


# ---- device-pointer forms (item 4) --------------------------------------------------------------------------------
@pytest.mark.parametrize("group", ["g1", "g2"])
def test_mul_sum_msm_dev_match_host_forms_and_oracle(eng, group):
    import torch
    pb = 96 if group == "g1" else 192
    rnd, ref_mul, ref_sum = (rand_g1, RC.g1_mul, RC.g1_sum) if group == "g1" else (rand_g2, RC.g2_mul, RC.g2_sum)
    host_mul, host_sum, host_msm = (eng.g1_mul_batch, eng.g1_sum, eng.g1_msm) if group == "g1" else (eng.g2_mul_batch, eng.g2_sum, eng.g2_msm)
    xs = P.XORShift(301)
    n = 70
    pts = [rnd(xs) for _ in range(n)]
    ks = [sk_bytes(xs) for _ in range(n)]
    ks[0] = bytes(32); ks[1] = (1).to_bytes(32, "big"); ks[2] = P.R_ORDER.to_bytes(32, "big"); ks[3] = (P.R_ORDER - 1).to_bytes(32, "big")
    pts[5] = bytes(pb)                                                     # the all-zero record: infinity in, infinity out
    dev = torch.device("cuda", 0)
    d_p, d_k = _dev(b"".join(pts)), _dev(b"".join(ks))
    d_out = torch.full((n * pb,), 0xAB, dtype=torch.uint8, device=dev); d_inf = torch.full((n,), 7, dtype=torch.uint8, device=dev)
    eng.mul_batch_dev(group, d_p.data_ptr(), d_k.data_ptr(), d_out.data_ptr(), d_inf.data_ptr(), n)
    got = d_out.cpu().numpy().reshape(n, pb); ginf = d_inf.cpu().numpy()
    h_out, h_inf = host_mul(b"".join(pts), b"".join(ks), n)
    assert np.array_equal(got, h_out) and np.array_equal(ginf.astype(bool), h_inf) and set(ginf.tolist()) <= {0, 1}
    for i in range(n):
        e = None if i == 5 else ref_mul(pts[i], ks[i])
        assert (e is None) == bool(ginf[i]) and (e is None or got[i].tobytes() == e), i
    # the generator as the common multiplicand (d_pts = NULL): PrivToPub
    eng.mul_batch_dev(group, 0, d_k.data_ptr(), d_out.data_ptr(), d_inf.data_ptr(), n)
    got = d_out.cpu().numpy().reshape(n, pb); ginf = d_inf.cpu().numpy()
    gen = RC.g1_generator() if group == "g1" else RC.g2_generator()
    for i in (0, 1, 2, 3, 4, 33, n - 1):
        e = ref_mul(gen, ks[i])
        assert (e is None) == bool(ginf[i]) and (e is None or got[i].tobytes() == e), i
    # on a caller-owned stream
    st = torch.cuda.Stream(device=dev)
    d_out2 = torch.zeros_like(d_out); d_inf2 = torch.zeros_like(d_inf)
    eng.mul_batch_dev(group, 0, d_k.data_ptr(), d_out2.data_ptr(), d_inf2.data_ptr(), n, stream=st.cuda_stream)
    assert torch.equal(d_out, d_out2) and torch.equal(d_inf, d_inf2)
    # sums: resident points, optional infinity flags
    finite = [p for i, p in enumerate(pts) if i != 5]
    d_f = _dev(b"".join(finite))
    d_one = torch.zeros(pb, dtype=torch.uint8, device=dev)
    for m in (1, 2, 3, 64, 65, len(finite)):
        inf = eng.sum_dev(group, d_f.data_ptr(), 0, m, d_one.data_ptr())
        assert not inf and d_one.cpu().numpy().tobytes() == ref_sum(b"".join(finite[:m]), m) == host_sum(b"".join(finite[:m]), m)
    flags = np.zeros(len(finite), dtype=np.uint8); flags[1] = 1; flags[40] = 1
    inf = eng.sum_dev(group, d_f.data_ptr(), _dev(flags).data_ptr(), len(finite), d_one.data_ptr())
    keep = [p for i, p in enumerate(finite) if not flags[i]]
    assert not inf and d_one.cpu().numpy().tobytes() == ref_sum(b"".join(keep), len(keep))
    assert eng.sum_dev(group, d_f.data_ptr(), 0, 0, d_one.data_ptr()) is True and not d_one.cpu().numpy().any()
    if group == "g1":
        d_pm = _dev(finite[0] + _neg_g1(finite[0]))
        assert eng.sum_dev(group, d_pm.data_ptr(), 0, 2, d_one.data_ptr()) is True
    # MSM below the bucket threshold (per-point multiples + tree sum on the device)
    for m in (1, 2, 7, n):
        inf = eng.msm_dev(group, d_p.data_ptr(), d_k.data_ptr(), m, d_one.data_ptr())
        e = [ref_mul(p, k) for i, (p, k) in enumerate(zip(pts[:m], ks[:m])) if i != 5]
        e = [x for x in e if x is not None]
        want = ref_sum(b"".join(e), len(e)) if e else None
        assert inf == (want is None) and (want is None or d_one.cpu().numpy().tobytes() == want), m
        assert host_msm(b"".join(pts[:m]), b"".join(ks[:m]), m) == want
    # a host pointer is refused, not dereferenced
    hbuf = np.zeros(pb, dtype=np.uint8)
    with pytest.raises(eng.BlsmiError):
        eng.msm_dev(group, d_p.data_ptr(), d_k.data_ptr(), 3, hbuf.ctypes.data)


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_dev_bucket_method_matches_host_form(group):
    """n = 2^17: the bucket method, inputs resident; the same point as the host form and as the scalar identity
    sum_i k_i (b_{i mod 64} G) = (sum_i k_i b_{i mod 64}) G evaluated by the oracle."""
    import torch
    from bls_amd import engine as eng
    eng.init(0)
    pb = 96 if group == "g1" else 192
    n = 1 << 17
    rng = np.random.default_rng(55)
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f
    base = 64
    bk = rng.integers(0, 256, size=(base, 32), dtype=np.uint8); bk[:, 0] &= 0x3f
    gen = RC.g1_generator() if group == "g1" else RC.g2_generator()
    bpts, _ = (eng.g1_mul_batch if group == "g1" else eng.g2_mul_batch)(gen * base, bk.reshape(-1), base)
    pts = np.tile(bpts, (n // base, 1))
    d_p, d_k = _dev(pts.reshape(-1)), _dev(k.reshape(-1))
    d_one = torch.zeros(pb, dtype=torch.uint8, device=torch.device("cuda", 0))
    assert eng.msm_dev(group, d_p.data_ptr(), d_k.data_ptr(), n, d_one.data_ptr()) is False
    got = d_one.cpu().numpy().tobytes()
    assert got == (eng.g1_msm if group == "g1" else eng.g2_msm)(pts.reshape(-1), k.reshape(-1), n)
    acc = 0
    kk = k.reshape(n // base, base, 32)
    for j in range(base):
        col = sum(int.from_bytes(kk[i, j].tobytes(), "big") for i in range(n // base))
        acc = (acc + int.from_bytes(bk[j].tobytes(), "big") * col) % P.R_ORDER
    assert got == (RC.g1_mul if group == "g1" else RC.g2_mul)(gen, acc.to_bytes(32, "big"))


def _aggregate_case(group, n, seed, nkeys=8):
    """n distinct messages signed by nkeys seeded keys (tiled), signatures made and summed on the device."""
    from bls_amd import engine as eng
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    xs = P.XORShift(seed)
    sks = [sk_bytes(xs) for _ in range(nkeys)]
    pks = [o.priv_to_pub(sk) for sk in sks]
    msgs = [b"aggregate %d of seed %d" % (i, seed) + bytes(i % 5) for i in range(n)]      # ragged lengths
    if group == "g2pubs":
        h = eng.hash_g1_batch(msgs); sigs, _ = eng.g1_mul_batch(h.reshape(-1), b"".join(sks[i % nkeys] for i in range(n)), n)
        agg = eng.g1_sum(sigs.reshape(-1), n)
    else:
        h = eng.hash_g2_batch(msgs); sigs, _ = eng.g2_mul_batch(h.reshape(-1), b"".join(sks[i % nkeys] for i in range(n)), n)
        agg = eng.g2_sum(sigs.reshape(-1), n)
    return msgs, [pks[i % nkeys] for i in range(n)], agg, o


@pytest.mark.parametrize("group", ["g2pubs", "g1pubs"])
@pytest.mark.parametrize("n", [9, 5000])
def test_verify_aggregate_dev_matches_host_form_and_oracle(eng, group, n):
    msgs, pks, agg, o = _aggregate_case(group, n, 400 + n)
    host = eng.g2pubs_verify_aggregate if group == "g2pubs" else eng.g1pubs_verify_aggregate

    def dev_call(ms, ks, sig):
        buf = np.frombuffer(b"".join(ms) or b"\0", dtype=np.uint8)
        off = np.zeros(len(ms) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(m) for m in ms])
        d_m, d_o, d_k = _dev(buf), _dev(off), _dev(b"".join(ks))
        return eng.verify_aggregate_dev(group, d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), sig, len(ms))

    assert dev_call(msgs, pks, agg) is True and host(msgs, b"".join(pks), agg) is True
    if n < 50:
        assert o.verify_aggregate(agg, pks, msgs) is True
    sw = list(pks); sw[n // 2], sw[n // 2 + 1] = sw[n // 2 + 1], sw[n // 2]
    assert dev_call(msgs, sw, agg) is False and host(msgs, b"".join(sw), agg) is False            # keys against the wrong messages
    dup = list(msgs); dup[n - 1] = dup[2]
    assert dev_call(dup, pks, agg) is False and host(dup, b"".join(pks), agg) is False             # duplicate message (g2pubs/bls.go:245-261)
    emp = list(msgs); emp[n // 3] = b""
    assert dev_call(emp, pks, agg) is False and host(emp, b"".join(pks), agg) is False             # empty message == nil upstream
    infk = list(pks); infk[n - 2] = bytes(len(pks[0]))
    assert dev_call(msgs, infk, agg) is False and host(msgs, b"".join(infk), agg) is False         # key at infinity: the reference panics
    assert dev_call(msgs, pks, bytes(len(agg))) is False                                            # signature at infinity
    # duplicates that a fingerprint-only screen would miss or over-report: equal prefixes, different lengths -> still distinct
    tricky = list(msgs); tricky[0] = b"abc"; tricky[1] = b"abc\0"; tricky[2] = b"abc\0\0"
    assert dev_call(tricky, pks, agg) is False                                                      # distinct messages, wrong signature: False by the pairing
    # ... and the screen itself: swap two messages AND their keys -> same multiset, still valid, nothing is "duplicate"
    perm = list(range(n)); perm[0], perm[n - 1] = perm[n - 1], perm[0]
    assert dev_call([msgs[i] for i in perm], [pks[i] for i in perm], agg) is True


def test_verify_aggregate_with_domain_large_and_dev(eng):
    """VerifyAggregateWithDomain at n >= 8192 (item 5c): no duplicate rejection in this function (g1pubs/bls.go:300-311), so a
    repeated message with a matching aggregate verifies; host form, device form and (on a small prefix) the oracle agree."""
    n = 8200
    dom = bytes(range(8))
    xs = P.XORShift(808)
    nk = 8
    sks = [sk_bytes(xs) for _ in range(nk)]
    pks = [RC.g1pubs.priv_to_pub(sk) for sk in sks]
    msgs = [hashlib.sha256(b"domain msg %d" % i).digest() for i in range(n)]
    msgs[n - 1] = msgs[0]                                                   # a repeat: allowed here
    h = eng.hash_g2_with_domain_batch(msgs, dom)
    assert h[7].tobytes() == RC.hash_g2_with_domain(msgs[7], dom)
    sigs, _ = eng.g2_mul_batch(h.reshape(-1), b"".join(sks[i % nk] for i in range(n)), n)
    agg = eng.g2_sum(sigs.reshape(-1), n)
    allpk = b"".join(pks[i % nk] for i in range(n))
    assert eng.g1pubs_verify_aggregate_with_domain(msgs, dom, allpk, agg) is True
    d_m, d_d, d_k = _dev(b"".join(msgs)), _dev(dom), _dev(allpk)
    assert eng.verify_aggregate_with_domain_dev(d_m.data_ptr(), d_d.data_ptr(), d_k.data_ptr(), agg, n) is True
    # VerifyWithDomain per tuple (g1pubs/bls.go:171-174), host form and device form: same verdict vector, wrong keys rejected
    import torch
    swapped = bytearray(allpk); swapped[96 * 17:96 * 18] = pks[(17 + 1) % nk]; swapped[96 * 8199:96 * 8200] = pks[(8199 + 1) % nk]
    want = np.ones(n, dtype=bool); want[17] = want[8199] = False
    assert np.array_equal(eng.g1pubs_verify_with_domain_batch(msgs, dom, bytes(swapped), sigs.reshape(-1)), want)
    d_ok = torch.zeros(n, dtype=torch.uint8, device=d_m.device)
    eng.g1pubs_verify_with_domain_batch_dev(d_m.data_ptr(), d_d.data_ptr(), _dev(bytes(swapped)).data_ptr(), _dev(sigs.reshape(-1)).data_ptr(), 0, d_ok.data_ptr(), n)
    assert np.array_equal(d_ok.cpu().numpy().astype(bool), want)
    bad = bytearray(allpk); bad[96 * 4000:96 * 4001] = pks[(4000 + 1) % nk]
    assert eng.g1pubs_verify_aggregate_with_domain(msgs, dom, bytes(bad), agg) is False
    assert eng.verify_aggregate_with_domain_dev(d_m.data_ptr(), d_d.data_ptr(), _dev(bytes(bad)).data_ptr(), agg, n) is False
    assert eng.g1pubs_verify_aggregate_with_domain(msgs, bytes(8), allpk, agg) is False          # another domain
    small = eng.g2_sum(sigs[:6].reshape(-1), 6)
    assert eng.g1pubs_verify_aggregate_with_domain(msgs[:6], dom, allpk[:96 * 6], small) is True
    assert RC.g1pubs.verify_aggregate_with_domain(small, [pks[i % nk] for i in range(6)], msgs[:6], dom) is True


# ---- non-canonical encodings (item 5a) ----------------------------------------------------------------------------
def test_noncanonical_field_encodings(eng):
    """x in [q, 2^381) under a valid compression header: FQReprToFQ gives 0 (fq.go:49-56), GetG1PointFromX(0) = (0, 2) or (0, q-2)
    -- on the curve, outside the subgroup -- so Deserialize* reports "not in correct subgroup" (g1.go:199-227) and the
    unchecked form returns that point.  Same for G2 with either coordinate of x out of range."""
    def enc1(x, greatest):
        b = bytearray(x.to_bytes(48, "big")); b[0] |= 0x80 | (0x20 if greatest else 0); return bytes(b)

    def enc2(c0, c1, greatest):
        b = bytearray(c1.to_bytes(48, "big") + c0.to_bytes(48, "big")); b[0] |= 0x80 | (0x20 if greatest else 0); return bytes(b)
    top = (1 << 381) - 1
    xs1 = [Q, Q + 1, Q + 12345, top, Q - 1, 0]
    c1 = [enc1(x, g) for x in xs1 for g in (False, True)]
    for check in (0, 1, 2):
        out, inf, err = eng.g1_decompress_batch(b"".join(c1), len(c1), check)
        for i, c in enumerate(c1):
            e, pt = RC.g1_decompress(c, checked=bool(check))
            assert int(err[i]) == e, (check, i, int(err[i]), e)
            if e == 0:
                assert out[i].tobytes() == pt and not inf[i], (check, i)
    # the x >= q cases land on x = 0: same output as the canonical encoding of 0
    out, _, err = eng.g1_decompress_batch(enc1(Q + 5, False) + enc1(0, False), 2, 0)
    assert not err.any() and out[0].tobytes() == out[1].tobytes() and int.from_bytes(out[0].tobytes()[:48], "big") == 0
    cases2 = [(Q, 1), (1, Q), (Q + 7, Q + 9), (top, 3), (3, top), (0, 0), (2, 5)]
    c2 = [enc2(a, b, g) for a, b in cases2 for g in (False, True)]
    for check in (0, 1, 2):
        out, inf, err = eng.g2_decompress_batch(b"".join(c2), len(c2), check)
        for i, c in enumerate(c2):
            e, pt = RC.g2_decompress(c, checked=bool(check))
            assert int(err[i]) == e, (check, i, int(err[i]), e)
            if e == 0:
                assert out[i].tobytes() == pt and not inf[i], (check, i)
    # Deserialize + Verify: a key or signature with x >= q fails to deserialise (subgroup check on) -> verdict 0 + error code
    xs = P.XORShift(66)
    sk = sk_bytes(xs); msg = b"non-canonical"
    pk, sig = RC.g2pubs.priv_to_pub(sk), RC.g2pubs.sign(msg, sk)
    cpk, csig = RC.g2_compress(pk), RC.g1_compress(sig)
    ok, epk, esg = eng.verify_serialized_batch("g2pubs", [msg] * 3, cpk + cpk + enc2(Q, 1, False), csig + enc1(Q + 1, True) + csig, True)
    assert list(ok) == [True, False, False] and int(epk[0]) == 0 and int(esg[1]) == RC.g1_decompress(enc1(Q + 1, True))[0] != 0 and int(epk[2]) == RC.g2_decompress(enc2(Q, 1, False))[0] != 0
    # affine inputs at the C ABI with a coordinate >= q: the device reads them as FQReprToFQ does (0), like the oracle
    big = (Q + 3).to_bytes(48, "big")
    weird_sig = big + sig[48:]
    assert eng.g2pubs_verify_batch([msg], pk, weird_sig)[0][0] == RC.g2pubs.verify(msg, pk, weird_sig)
    g2 = rand_g2(xs)
    weird_p = big + (2).to_bytes(48, "big")                                # reads as (0, 2): on the curve
    assert np.array_equal(eng.pairing_batch(weird_p, g2, 1), RC.pairing_batch(weird_p, g2, 1))
    assert np.array_equal(eng.pairing_batch(weird_p, g2, 1), eng.pairing_batch(bytes(48) + (2).to_bytes(48, "big"), g2, 1))


# ---- on-curve points outside the subgroup (item 5b) ---------------------------------------------------------------
def _torsion_points():
    g1s, g2s = [], []
    x = 0
    while len(g1s) < 6:
        pt = P.g1_from_x(x, bool(x & 1))
        if pt is not None and not P.g1_in_subgroup(pt):
            g1s.append(RC.g1_decompress(P.g1_compress(pt), checked=False)[1])
        x += 1
    x0 = 0
    while len(g2s) < 6:
        pt = P.g2_from_x((x0 % 7, x0 // 7 + 1), bool(x0 & 1))
        if pt is not None and not P.g2_in_subgroup(pt):
            g2s.append(RC.g2_decompress(P.g2_compress(pt), checked=False)[1])
        x0 += 1
    return g1s, g2s


def test_points_outside_the_subgroup_through_pairing_and_verify(eng):
    g1s, g2s = _torsion_points()
    xs = P.XORShift(515)
    sub1 = [rand_g1(xs) for _ in range(6)]; sub2 = [rand_g2(xs) for _ in range(6)]
    # every combination class: (torsion, subgroup), (subgroup, torsion), (torsion, torsion)
    a = g1s + sub1 + g1s
    b = sub2 + g2s + g2s
    n = len(a)
    want = RC.pairing_batch(b"".join(a), b"".join(b), n)
    got = eng.pairing_batch(b"".join(a), b"".join(b), n)
    assert np.array_equal(got, want)
    # the exported MillerLoop keeps the reference's steps: its value matches as well
    ml = eng.miller_loop_batch(b"".join(a), b"".join(b), n)
    for i in range(n):
        assert bool(np.array_equal(ml[i], RC.miller_loop(a[i], b[i], 1))), i
    # verify: keys / signatures that are curve points outside the subgroup get the oracle's verdict (both packages)
    sk = sk_bytes(xs); msg = b"outside the subgroup"
    pk2, sig1 = RC.g2pubs.priv_to_pub(sk), RC.g2pubs.sign(msg, sk)
    msgs = [msg] * 5
    pks = [pk2, g2s[0], pk2, g2s[1], g2s[2]]
    sigs = [sig1, sig1, g1s[0], g1s[1], g1s[2]]
    ok, _ = eng.g2pubs_verify_batch(msgs, b"".join(pks), b"".join(sigs))
    assert list(ok) == [RC.g2pubs.verify(m, p, s) for m, p, s in zip(msgs, pks, sigs)] and ok[0]
    pk1, sig2 = RC.g1pubs.priv_to_pub(sk), RC.g1pubs.sign(msg, sk)
    pks = [pk1, g1s[3], pk1, g1s[4]]
    sigs = [sig2, sig2, g2s[3], g2s[4]]
    ok, _ = eng.g1pubs_verify_batch(msgs[:4], b"".join(pks), b"".join(sigs))
    assert list(ok) == [RC.g1pubs.verify(m, p, s) for m, p, s in zip(msgs, pks, sigs)] and ok[0]
    # a small aggregate with a torsion key: the product path (miller1raw + aggtail / k_miller1x2_pair) agrees with the oracle
    ms = [b"agg torsion %d" % i for i in range(3)]
    keys = [pk2, g2s[4], pk2]
    agg = RC.g1_sum(b"".join(RC.g2pubs.sign(m, sk) for m in ms), 3)
    assert eng.g2pubs_verify_aggregate(ms, b"".join(keys), agg) == RC.g2pubs.verify_aggregate(agg, keys, ms)


# ---- a slice of the differential soak (item 5d) -------------------------------------------------------------------
def test_soak_slice_latency_programs_against_throughput_kernels():
    """Seeded, ~20 s: random tuples through the latency programs (one tuple per wave) and the throughput kernels (one
    per lane pair) must agree byte for byte, and a sample of each must equal the oracle.  tools/soak.py is the long form."""
    from bls_amd import engine
    engine.init(0)
    rng = np.random.default_rng(20240928)

    def scal(m):
        raw = rng.integers(0, 256, size=(m, 32), dtype=np.uint8); raw[:, 0] &= 0x3f; raw[:, 31] |= 1
        return raw
    n = 4096
    try:
        g1, _ = engine.g1_mul_batch(RC.g1_generator() * n, scal(n).reshape(-1), n)
        g2, _ = engine.g2_mul_batch(RC.g2_generator() * n, scal(n).reshape(-1), n)
        outs = {}
        for thr in (0, 4096):
            engine.set_latency_threshold(thr)
            outs[thr] = engine.pairing_batch(g1.reshape(-1), g2.reshape(-1), n)
        assert np.array_equal(outs[0], outs[4096]), "pairing: the two paths differ"
        for i in rng.integers(0, n, size=24):
            assert np.array_equal(outs[0][i], RC.pairing_batch(g1[i].tobytes(), g2[i].tobytes(), 1)[0]), i
        msgs = [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in rng.integers(0, 200, size=1500)]
        for fn, ref in ((engine.hash_g1_batch, RC.hash_g1), (engine.hash_g2_batch, RC.hash_g2)):
            engine.set_latency_threshold(8192); x = fn(msgs)
            engine.set_latency_threshold(0); y = fn(msgs)
            assert np.array_equal(x, y)
            for i in rng.integers(0, len(msgs), size=12):
                assert x[i].tobytes() == ref(msgs[i]), i
        m = 1500
        k = scal(m); k[::7, :20] = 0; k[::11] = 0
        for mul, pts, ref in ((engine.g1_mul_batch, g1[:m], RC.g1_mul), (engine.g2_mul_batch, g2[:m], RC.g2_mul)):
            engine.set_latency_threshold(8192); x, ix = mul(pts.reshape(-1), k.reshape(-1), m)
            engine.set_latency_threshold(0); y, iy = mul(pts.reshape(-1), k.reshape(-1), m)
            assert np.array_equal(x, y) and np.array_equal(ix, iy)
            assert ix[::11].all()
            for i in rng.integers(0, m, size=12):
                e = ref(pts[i].tobytes(), k[i].tobytes())
                assert (e is None and ix[i]) or x[i].tobytes() == e, i
        # verify verdicts, every 5th tuple corrupted, both paths
        nv = 1200
        sks = scal(64)
        pk, _ = engine.g2_mul_batch(RC.g2_generator() * 64, sks.reshape(-1), 64)
        vm = [b"soak %d" % i for i in range(nv)]
        h = engine.hash_g1_batch(vm)
        sg, _ = engine.g1_mul_batch(h.reshape(-1), np.tile(sks, (nv // 64 + 1, 1))[:nv].reshape(-1), nv)
        pks = np.tile(pk, (nv // 64 + 1, 1))[:nv].copy()
        expect = np.ones(nv, dtype=bool)
        for i in range(4, nv, 5):
            pks[i] = pk[(i + 1) % 64]; expect[i] = False
        for thr in (4096, 0):
            engine.set_latency_threshold(thr)
            ok, _ = engine.g2pubs_verify_batch(vm, pks.reshape(-1), sg.reshape(-1))
            assert np.array_equal(ok, expect), thr
        for i in (0, 4, 5, 9, nv - 1):
            assert RC.g2pubs.verify(vm[i], pks[i].tobytes(), sg[i].tobytes()) == bool(expect[i])
    finally:
        engine.set_latency_threshold(8192)


# ---- every shard size the 8-GPU run will see, on this one GPU (item 2) --------------------------------------------
def _worker(env_extra, args):
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shard_worker8.py")] + args, env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("SHARD8_RESULT ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(line[-1][len("SHARD8_RESULT "):])


@pytest.mark.parametrize("rccl", ["0", "1"])
def test_eight_logical_shards_full_size_on_one_gpu(rccl):
    """BLSMI_SHARDS=8 on this one GPU: the 2^20-signature VerifyAggregate (8 x 131 072, partial products gathered) and
    8 x 65 536 verifies with the bitmap (all-reduce) -- the shard sizes, slot arithmetic and merges of the 8-GPU run.
    rccl=1 routes the exchanges through a real (one-rank) RCCL communicator."""
    res = _worker({"BLSMI_SHARDS": "8", "BLSMI_FORCE_RCCL": rccl}, [])
    assert res["shards"] == 8 and res["ok"], res


# ---- endomorphism scalar multiplication: edge scalars, both paths, and the opt-out for arbitrary curve points ---------------------
def test_endomorphism_scalar_multiplication_edge_scalars(eng):
    """The decompositions behind the default scalar multiplication (bls_amd/csrc/glv.cuh, glv_model.py) are integer identities for
    EVERY 256-bit scalar: the boundaries of the divisions (multiples of z and z^2, r, 2^256 - 1 ...) against the oracle's bit-serial
    MulFR (g1.go:80-90, g2.go:92-102), on the latency programs and the throughput ladders, for points, the generators (fixed-base
    tables) and the MSM (whose digit pass reduces mod r first)."""
    Z = P.BLS_X; Z2 = Z * Z; R = P.R_ORDER
    ks = [0, 1, 2, Z - 1, Z, Z + 1, Z2 - 1, Z2, Z2 + 1, Z**3 - 1, Z**3, Z**3 + 1, R - 1, R, R + 1, 2 * R, (1 << 128) - 1, 1 << 128, (1 << 129) - 1,
          (1 << 255) - 1, 1 << 255, (1 << 256) - 1, (1 << 256) - Z2, 0x0101010101010101010101010101010101010101010101010101010101010101 ]
    xs = P.XORShift(909)
    ks += [P.rand_int(xs, 1 << 256) for _ in range(11)]
    n = len(ks)
    kb = b"".join(k.to_bytes(32, "big") for k in ks)
    p1 = [rand_g1(xs) for _ in range(n)]; p2 = [rand_g2(xs) for _ in range(n)]
    for pts, mul, gen_mul, msm, ref_mul, ref_sum, gen in ((p1, eng.g1_mul_batch, eng.g1_mul_generator_batch, eng.g1_msm, RC.g1_mul, RC.g1_sum, RC.g1_generator()),
                                                          (p2, eng.g2_mul_batch, eng.g2_mul_generator_batch, eng.g2_msm, RC.g2_mul, RC.g2_sum, RC.g2_generator())):
        out, inf = mul(b"".join(pts), kb, n)
        want = [ref_mul(p, k.to_bytes(32, "big")) for p, k in zip(pts, ks)]
        for i in range(n):
            assert (want[i] is None) == bool(inf[i]) and (want[i] is None or out[i].tobytes() == want[i]), (i, hex(ks[i]))
        out, inf = gen_mul(kb, n)
        for i in range(n):
            e = ref_mul(gen, ks[i].to_bytes(32, "big"))
            assert (e is None) == bool(inf[i]) and (e is None or out[i].tobytes() == e), (i, hex(ks[i]))
        fin = [w for w in want if w is not None]
        assert msm(b"".join(pts), kb, n) == ref_sum(b"".join(fin), len(fin))


def test_scalar_multiplication_of_arbitrary_curve_points_needs_the_opt_out(eng):
    """For a curve point OUTSIDE the subgroup the endomorphisms are not multiplications, so the default ladder is not MulFR there;
    blsmi_set_mul_assume_subgroup(0) selects the plain windowed ladder, which is -- for every curve point (g1.go:80-90)."""
    g1s, g2s = _torsion_points()
    xs = P.XORShift(910)
    ks = [sk_bytes(xs) for _ in range(4)]
    try:
        eng.set_mul_assume_subgroup(False)
        out, inf = eng.g1_mul_batch(b"".join(g1s[:4]), b"".join(ks), 4)
        for i in range(4):
            assert out[i].tobytes() == RC.g1_mul(g1s[i], ks[i]) and not inf[i]
        out, inf = eng.g2_mul_batch(b"".join(g2s[:4]), b"".join(ks), 4)
        for i in range(4):
            assert out[i].tobytes() == RC.g2_mul(g2s[i], ks[i]) and not inf[i]
        # subgroup points: both modes agree with the oracle
        p = rand_g1(xs)
        assert eng.g1_mul_batch(p, ks[0], 1)[0][0].tobytes() == RC.g1_mul(p, ks[0])
    finally:
        eng.set_mul_assume_subgroup(True)
    p = rand_g1(xs)
    assert eng.g1_mul_batch(p, ks[0], 1)[0][0].tobytes() == RC.g1_mul(p, ks[0])


def test_msm_bucket_method_with_scalars_beyond_r_and_points_at_infinity():
    """n = 2^17 (the bucket method over decomposed scalars): scalars up to 2^256 - 1 (the digit pass reduces them mod r first), a few
    all-zero records (infinity) among the points; the result equals the oracle's multiple of the generator for the folded scalar."""
    import torch
    from bls_amd import engine as eng
    eng.init(0)
    n = 1 << 17
    rng = np.random.default_rng(77)
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)                      # full 256-bit scalars
    base = 32
    bk = rng.integers(0, 256, size=(base, 32), dtype=np.uint8); bk[:, 0] &= 0x3f
    for grp, pb, gen, ref_mul in (("g1", 96, RC.g1_generator(), RC.g1_mul), ("g2", 192, RC.g2_generator(), RC.g2_mul)):
        bpts, _ = (eng.g1_mul_generator_batch if grp == "g1" else eng.g2_mul_generator_batch)(bk.reshape(-1), base)
        pts = np.tile(bpts, (n // base, 1)).copy()
        holes = [5, 70000, n - 1]
        for h in holes:
            pts[h] = 0                                                             # the all-zero record: the point at infinity
        got = (eng.g1_msm if grp == "g1" else eng.g2_msm)(pts.reshape(-1), k.reshape(-1), n)
        acc = 0
        kk = k.reshape(n // base, base, 32)
        skip = {(h // base, h % base) for h in holes}
        for j in range(base):
            col = sum(int.from_bytes(kk[i, j].tobytes(), "big") for i in range(n // base) if (i, j) not in skip)
            acc = (acc + int.from_bytes(bk[j].tobytes(), "big") * col) % P.R_ORDER
        assert got == ref_mul(gen, acc.to_bytes(32, "big")), grp


@pytest.mark.parametrize("force", ["", "1"])
def test_device_duplicate_table_and_its_fallback(force):
    """Duplicate-message rejection of VerifyAggregate for n > 4096 (g2pubs/bls.go:245-261): the device's keyed open-addressing table
    (k_util.hip) and -- BLSMI_DUP_FORCE_SORT=1 -- the path it takes when a probe sequence grows too long: the reference's sort on the
    host.  Same verdicts for a clean set, one duplicate, an empty message and n equal messages, resident and host buffers."""
    env = dict(os.environ)
    env.pop("BLSMI_DUP_FORCE_SORT", None)
    if force:
        env["BLSMI_DUP_FORCE_SORT"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dup_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("DUP_RESULT ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    res = json.loads(line[-1][len("DUP_RESULT "):])
    assert res["forced"] == bool(force) and res["ok"], res


@pytest.mark.parametrize("group", ["g1", "g2"])
def test_msm_with_a_crowded_bucket_on_both_grouping_paths(group):
    """The MSM groups its 16 n (bucket, point) items with a device radix sort (default) or, with blsmi_set_option("msm_sort", 0) / BLSMI_MSM_SORT=0 at start-up, with the exact histogram + scan +
    atomic scatter.  300 EQUAL scalars put 300 items into the same bucket of every window (mean population 4 at n = 2^17; below the skew limit of
    2 048): both paths must give the point the oracle gives through the scalar identity."""
    from bls_amd import engine as eng
    import ctypes
    eng.init(0)
    n = 1 << 17
    rng = np.random.default_rng(56)
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f
    k[1000:1300] = k[999]                                                  # 301 equal scalars
    base = 64
    bk = rng.integers(0, 256, size=(base, 32), dtype=np.uint8); bk[:, 0] &= 0x3f
    gen = RC.g1_generator() if group == "g1" else RC.g2_generator()
    bpts, _ = (eng.g1_mul_batch if group == "g1" else eng.g2_mul_batch)(gen * base, bk.reshape(-1), base)
    pts = np.tile(bpts, (n // base, 1))
    acc = 0
    kk = k.reshape(n // base, base, 32)
    for j in range(base):
        col = sum(int.from_bytes(kk[i, j].tobytes(), "big") for i in range(n // base))
        acc = (acc + int.from_bytes(bk[j].tobytes(), "big") * col) % P.R_ORDER
    want = (RC.g1_mul if group == "g1" else RC.g2_mul)(gen, acc.to_bytes(32, "big"))
    lib = eng._lib()
    try:
        for mode, marks in ((1, ("rocprim:radix_sort", "k_msm_runs")), (0, ("k_msm_hist_glv", "k_msm_scan", "k_msm_scatter_glv"))):
            eng.set_option("msm_sort", mode)                                   # an atomic the library reads per call (blsmi.hip: msm_bucket_glv_dev); the environment is read at start-up only
            lib.blsmi_set_profiling(1)
            got = (eng.g1_msm if group == "g1" else eng.g2_msm)(pts.reshape(-1), k.reshape(-1), n)
            buf = ctypes.create_string_buffer(8192)
            lib.blsmi_last_profile(buf, ctypes.c_size_t(8192)); lib.blsmi_set_profiling(0)
            names = buf.value.decode()
            assert all(m in names for m in marks), (mode, names)
            assert got == want, mode
    finally:
        eng.set_option("msm_sort", 1)


def test_page_locked_host_buffers(eng):
    """blsmi_host_alloc / blsmi_host_free: buffers the host entry points copy from and to by DMA; same values as from pageable memory"""
    n = 300
    g1, _ = eng.g1_mul_generator_batch(b"".join((i + 2).to_bytes(32, "big") for i in range(n)), n)
    g2, _ = eng.g2_mul_generator_batch(b"".join((3 * i + 5).to_bytes(32, "big") for i in range(n)), n)
    want = eng.pairing_batch(g1.reshape(-1), g2.reshape(-1), n)
    a, b, o = eng.HostBuffer(96 * n), eng.HostBuffer(192 * n), eng.HostBuffer(576 * n)
    a.a[:] = g1.reshape(-1); b.a[:] = g2.reshape(-1)
    got = eng.pairing_batch(a.a, b.a, n, out=o.a.view(np.uint64).reshape(n, 72))
    assert np.array_equal(got, want)
    assert got[0].tobytes() == RC.pairing_batch(g1[0].tobytes(), g2[0].tobytes(), 1)[0].tobytes()
    del got
    for h in (a, b, o):
        h.free(); h.free()                                                 # the second call is a no-op
    empty = eng.HostBuffer(0); empty.free()
