"""-m gpu, round 4: per-call ladder choice for scalar multiplication of arbitrary curve points, blsmi_trim and the arena's
retention cap, the latency hint, and the mid-size layout boundaries."""
import threading

import numpy as np
import pytest

from gpu_common import P, RC, rand_g1, rand_g2, sk_bytes
from test_gpu_round3 import _torsion_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from bls_amd import engine
    engine.init(0)
    return engine


def test_any_point_flag_selects_the_plain_ladder_per_call(eng):
    """MulFR accepts any curve point (g1.go:80-90, g2.go:92-102).  any_point=True gives the oracle's multiple for points outside the
    subgroup, on the small-call path, on the throughput kernels and in the MSM -- while default-ladder calls run CONCURRENTLY on other
    threads with their own (subgroup) inputs and keep their results: the choice is per call, no process-wide state is touched."""
    g1s, g2s = _torsion_points()
    xs = P.XORShift(4101)
    ks = [sk_bytes(xs) for _ in range(6)]
    want1 = [RC.g1_mul(p, k) for p, k in zip(g1s, ks)]
    want2 = [RC.g2_mul(p, k) for p, k in zip(g2s, ks)]
    sub1 = [rand_g1(xs) for _ in range(6)]; sub2 = [rand_g2(xs) for _ in range(6)]
    wsub1 = [RC.g1_mul(p, k) for p, k in zip(sub1, ks)]; wsub2 = [RC.g2_mul(p, k) for p, k in zip(sub2, ks)]
    stop = threading.Event()
    bad = []

    def default_caller():
        while not stop.is_set():
            o1, _ = eng.g1_mul_batch(b"".join(sub1), b"".join(ks), 6)
            o2, _ = eng.g2_mul_batch(b"".join(sub2), b"".join(ks), 6)
            if [x.tobytes() for x in o1] != wsub1 or [x.tobytes() for x in o2] != wsub2:
                bad.append("default ladder disturbed")
    th = [threading.Thread(target=default_caller) for _ in range(2)]
    for t in th:
        t.start()
    try:
        for lat in (8192, 0):                                              # one multiplication per wave / the throughput ladders
            eng.set_latency_threshold(lat)
            for reps in (1, 40):                                           # 6 and 240 points (whole and ragged workgroups)
                o1, i1 = eng.g1_mul_batch(b"".join(g1s) * reps, b"".join(ks) * reps, 6 * reps, any_point=True)
                o2, i2 = eng.g2_mul_batch(b"".join(g2s) * reps, b"".join(ks) * reps, 6 * reps, any_point=True)
                assert [x.tobytes() for x in o1] == want1 * reps and not i1.any()
                assert [x.tobytes() for x in o2] == want2 * reps and not i2.any()
                # subgroup points through the any-point ladder: the same bytes as the default one
                o1, _ = eng.g1_mul_batch(b"".join(sub1), b"".join(ks), 6, any_point=True)
                assert [x.tobytes() for x in o1] == wsub1
    finally:
        eng.set_latency_threshold(8192)
        stop.set()
        for t in th:
            t.join()
    assert not bad
    # MSM over points outside the subgroup == sum of the oracle's multiples (small n: multiples + tree sum; the default would be wrong here)
    assert eng.g1_msm(b"".join(g1s), b"".join(ks), 6, any_point=True) == RC.g1_sum(b"".join(want1), 6)
    assert eng.g2_msm(b"".join(g2s), b"".join(ks), 6, any_point=True) == RC.g2_sum(b"".join(want2), 6)
    # the default ladder on these points is NOT MulFR: that is the documented precondition, pinned here so that it stays documented
    o1, _ = eng.g1_mul_batch(b"".join(g1s), b"".join(ks), 6)
    assert [x.tobytes() for x in o1] != want1


def test_any_point_device_forms_and_bucket_msm(eng):
    """the *_dev_ex forms, and the bucket MSM (n = 2^17) through the plain buckets when any_point is set: a batch that contains points
    outside the subgroup against sum-of-multiples computed by the any-point ladder itself (cross-check) and a sample against the oracle"""
    import torch
    dev = torch.device("cuda", 0)
    g1s, g2s = _torsion_points()
    n = 1 << 17
    rng = np.random.default_rng(4102)
    k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f
    xs = P.XORShift(4103)
    for grp, pb, tors, rnd, ref_mul, summ in (("g1", 96, g1s, rand_g1, RC.g1_mul, eng.g1_sum), ("g2", 192, g2s, rand_g2, RC.g2_mul, eng.g2_sum)):
        base = [np.frombuffer(t, dtype=np.uint8) for t in tors] + [np.frombuffer(rnd(xs), dtype=np.uint8) for _ in range(10)]
        pts = np.ascontiguousarray(np.tile(np.stack(base), (n // 16, 1)))
        d_p = torch.from_numpy(pts.reshape(-1)).to(dev); d_k = torch.from_numpy(k.reshape(-1)).to(dev)
        d_out = torch.empty(n * pb, dtype=torch.uint8, device=dev); d_inf = torch.empty(n, dtype=torch.uint8, device=dev)
        eng.mul_batch_dev(grp, d_p.data_ptr(), d_k.data_ptr(), d_out.data_ptr(), d_inf.data_ptr(), n, any_point=True)
        out = d_out.cpu().numpy().reshape(n, pb)
        inf = d_inf.cpu().numpy()
        # (a curve point outside the subgroup may have small order -- (0, 2) on E has order 3 -- so some multiples ARE the point at
        # infinity: the flag must agree with the oracle, which returns None there)
        for i in list(range(48)) + [n - 16, n - 11, n - 1]:
            want = ref_mul(pts[i].tobytes(), k[i].tobytes())
            assert bool(inf[i]) == (want is None), (grp, i)
            if want is not None:
                assert out[i].tobytes() == want, (grp, i)
        total = summ(out.reshape(-1), n, inf)
        d_one = torch.zeros(pb, dtype=torch.uint8, device=dev)
        assert eng.msm_dev(grp, d_p.data_ptr(), d_k.data_ptr(), n, d_one.data_ptr(), any_point=True) is False
        assert d_one.cpu().numpy().tobytes() == total, grp
        assert (eng.g1_msm if grp == "g1" else eng.g2_msm)(pts.reshape(-1), k.reshape(-1), n, any_point=True) == total
    lib = __import__("bls_amd._native", fromlist=["load"]).load()
    assert lib.blsmi_g1_mul_batch_ex(None, None, None, None, 0, 2) == -3           # unknown flag bits are refused


def test_trim_returns_the_temporaries_and_calls_still_work(eng):
    """blsmi_trim / blsmi_held_bytes: after a large call the contexts hold its temporaries (that is the design: steady state makes no
    allocator call); trim(0) gives all of it back, and the next call -- which has to allocate again -- returns the same results."""
    xs = P.XORShift(4104)
    n = 20000                                                              # above the latency hand-over: the throughput kernels' workspace
    g1 = rand_g1(xs) * 1; g2 = rand_g2(xs) * 1
    a = g1 * n; b = g2 * n
    want = RC.pairing_batch(g1, g2, 1)[0]
    out = eng.pairing_batch(a, b, n)
    assert np.array_equal(out[0], want) and np.array_equal(out[n - 1], want)
    held = eng.held_bytes()
    assert held >= 720 * n                                                 # at least the Miller-loop -> final-exponentiation hand-off
    freed = eng.trim(0)
    assert freed >= 720 * n and eng.held_bytes() == 0
    out = eng.pairing_batch(a, b, n)
    assert np.array_equal(out[0], want) and np.array_equal(out[n - 1], want)
    # keep_bytes is a per-context ceiling, not a request to free everything
    held = eng.held_bytes()
    assert eng.trim(1 << 40) == 0 and eng.held_bytes() == held


def test_retention_cap_releases_an_outsized_calls_temporaries():
    """BLSMI_ARENA_KEEP_MB=8: a call whose temporaries exceed the cap returns them when it ends (own process: read at initialisation)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from bls_amd import engine as e\n"
            "from gpu_common import RC\n"
            "e.init(0)\n"
            "n = 1 << 17\n"
            "k = np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x3f\n"
            "p, _ = e.g1_mul_generator_batch(k.reshape(-1), n)\n"
            "r1 = e.g1_msm(p.reshape(-1), k.reshape(-1), n)\n"
            "h1 = e.held_bytes()\n"
            "r2 = e.g1_msm(p.reshape(-1), k.reshape(-1), n)\n"
            "h2 = e.held_bytes()\n"
            "assert r1 == r2 and r1 == e.g1_sum(e.g1_mul_batch(p.reshape(-1), k.reshape(-1), n)[0].reshape(-1), n)\n"
            "assert h1 <= 8 << 20 and h2 <= 8 << 20, (h1, h2)\n"
            "print('OK', h1, h2)\n" % (root, os.path.join(root, "tests")))
    env = dict(os.environ); env["BLSMI_ARENA_KEEP_MB"] = "8"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


def test_latency_hint(eng):
    """blsmi_prefer_cpu: the lone calls that lose to one CPU core (BLSSign, G2Prepare, MillerLoop) are flagged, the ones that
    win are not, and the hint flips at the measured break-even"""
    E = eng
    assert E.prefer_cpu(E.SHAPE_SIGN, 1) and E.prefer_cpu(E.SHAPE_SIGN, 3) and not E.prefer_cpu(E.SHAPE_SIGN, 4)
    assert E.prefer_cpu(E.SHAPE_G2_PREPARE, 7) and not E.prefer_cpu(E.SHAPE_G2_PREPARE, 8)
    assert E.prefer_cpu(E.SHAPE_MILLER_LOOP, 1) and not E.prefer_cpu(E.SHAPE_MILLER_LOOP, 2)
    for shape in (E.SHAPE_PAIRING, E.SHAPE_FINAL_EXP, E.SHAPE_VERIFY, E.SHAPE_VERIFY_DOMAIN):
        assert not E.prefer_cpu(shape, 1)
    assert E.prefer_cpu(E.SHAPE_POINT_ADD, 38) and not E.prefer_cpu(E.SHAPE_POINT_ADD, 39)
    assert not E.prefer_cpu(99, 1) and not E.prefer_cpu(-1, 1)


def test_verify_aggregate_common_with_resident_keys(eng):
    """VerifyAggregateCommon (g2pubs/bls.go:275-278, g1pubs/bls.go:287-297 incl. the *WithDomain form) with the public keys resident in
    HBM: verdicts equal the oracle's and the host form's -- true aggregate, one key dropped from the signature, a foreign key in the
    set, empty message, n = 0 -- for a handful of signers and for 3 000 (tree sum over several levels)."""
    import torch
    dev = torch.device("cuda", 0)
    xs = P.XORShift(4105)
    msg = b"common message"; m32 = bytes(range(32)); dom = bytes(range(8))
    for group, O, nk in (("g2pubs", RC.g2pubs, 5), ("g1pubs", RC.g1pubs, 5), ("g2pubs", RC.g2pubs, 3000), ("g1pubs", RC.g1pubs, 3000)):
        sks = [sk_bytes(xs) for _ in range(7)]
        pks7 = [O.priv_to_pub(s) for s in sks]
        pkb = len(pks7[0])
        # nk signers: the 7 keys repeated (the aggregate signature is the matching multiple of the 7 signatures' sum)
        reps = [nk // 7 + (1 if i < nk % 7 else 0) for i in range(7)]
        pks = [pks7[i % 7] for i in range(nk)]
        mul = eng.g1_mul_batch if group == "g2pubs" else eng.g2_mul_batch
        summ = eng.g1_sum if group == "g2pubs" else eng.g2_sum
        def aggregate(message, sign, count):
            sigs = [sign(message, s) for s in sks]
            sb = len(sigs[0])
            scaled, inf = mul(b"".join(sigs), b"".join(int(c).to_bytes(32, "big") for c in count), 7)
            return summ(scaled.reshape(-1), 7, inf.astype(np.uint8))
        d_k = torch.from_numpy(np.frombuffer(b"".join(pks), dtype=np.uint8).copy()).to(dev)
        host = eng.g2pubs_verify_aggregate_common if group == "g2pubs" else eng.g1pubs_verify_aggregate_common
        for message in (msg, b""):
            agg = aggregate(message, O.sign, reps)
            assert eng.verify_aggregate_common_dev(group, d_k.data_ptr(), nk, message, agg) is True
            assert host(message, b"".join(pks), agg, nk) is True
            if nk <= 8:
                assert O.verify_aggregate_common(agg, pks, message) is True
            short = aggregate(message, O.sign, [reps[0] - 1] + reps[1:])                 # one signature missing
            assert eng.verify_aggregate_common_dev(group, d_k.data_ptr(), nk, message, short) is False
            if nk <= 8:
                assert O.verify_aggregate_common(short, pks, message) is False
        bad = d_k.clone(); bad[pkb * (nk - 1):pkb * nk] = torch.from_numpy(np.frombuffer(O.priv_to_pub(sk_bytes(xs)), dtype=np.uint8).copy()).to(dev)
        assert eng.verify_aggregate_common_dev(group, bad.data_ptr(), nk, msg, aggregate(msg, O.sign, reps)) is False
        assert eng.verify_aggregate_common_dev(group, 0, 0, msg, aggregate(msg, O.sign, reps)) is False       # no keys: the sum is infinity
        if group == "g1pubs":
            aggd = aggregate(m32, lambda m, s: RC.g1pubs.sign_with_domain(m, s, dom), reps)
            assert eng.verify_aggregate_common_dev(group, d_k.data_ptr(), nk, m32, aggd, domain=dom) is True
            assert eng.verify_aggregate_common_dev(group, d_k.data_ptr(), nk, m32, aggd, domain=bytes(8)) is False
            if nk <= 8:
                assert RC.g1pubs.verify_aggregate_common_with_domain(aggd, pks, m32, dom) is True


@pytest.mark.parametrize("group", ["g2pubs", "g1pubs"])
def test_small_verify_calls_with_the_signature_side_on_a_side_stream(eng, group):
    """Calls of up to 48 (g2pubs) / 320 (g1pubs) tuples run the signature side's Miller loop on a side stream beside the hash (programs miller1rawn + verify1s), larger
    latency-path calls the two-pair program (verify2): the verdict table of g2pubs/bls.go:159-162 / g1pubs/bls.go:165-168 -- wrong message, wrong
    key, negated signature, flagged and all-zero (infinity) records -- is the oracle's on both sides of the boundary, from host buffers and
    from resident ones, and a call's profile names the program it took."""
    import ctypes
    import torch
    from test_gpu_verify import _tuples
    msgs, pks, sigs, expect = _tuples(group, 66, 77)
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    for i in (0, 3, 7, 11):
        assert o.verify(msgs[i], pks[i], sigs[i]) == expect[i]
    fn = eng.g2pubs_verify_batch if group == "g2pubs" else eng.g1pubs_verify_batch
    sb = len(sigs[0])
    lib = eng._lib()
    lim = 48 if group == "g2pubs" else 320
    big = _tuples(group, 322, 78) if group == "g1pubs" else None
    cases = [(1, "k_lat:verify1s"), (2, "k_lat:verify1s"), (48, "k_lat:verify1s"), (49 if lim == 48 else 66, "k_lat:verify2" if lim == 48 else "k_lat:verify1s")]
    if big:
        got, _ = fn(big[0][:320], b"".join(big[1][:320]), b"".join(big[2][:320])); assert list(got) == big[3][:320]
        got, _ = fn(big[0][:321], b"".join(big[1][:321]), b"".join(big[2][:321])); assert list(got) == big[3][:321]
    for n, prog in cases:
        lib.blsmi_set_profiling(1)
        ok, _ = fn(msgs[:n], b"".join(pks[:n]), b"".join(sigs[:n]))
        buf = ctypes.create_string_buffer(4096); lib.blsmi_last_profile(buf, ctypes.c_size_t(4096)); lib.blsmi_set_profiling(0)
        assert prog in buf.value.decode(), (n, buf.value)
        assert list(ok) == expect[:n], n
        # a flagged tuple, and a signature given as the all-zero record: verdict 0, the neighbours keep theirs
        flags = [0] * n; flags[0] = 2
        s2 = list(sigs[:n])
        if n > 1: s2[1] = bytes(sb)
        ok, _ = fn(msgs[:n], b"".join(pks[:n]), b"".join(s2), flags)
        assert not ok[0] and (n == 1 or not ok[1]) and list(ok[2:]) == expect[2:n], n
    # resident inputs on the caller's own stream (the side stream waits for that stream's work on the signatures)
    dev = torch.device("cuda", 0)
    for n in (5, 48, 49, 66):
        mbuf, moff = eng._msgs(msgs[:n])
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            d_m = torch.from_numpy(mbuf.copy()).to(dev, non_blocking=True)
            d_o = torch.from_numpy(moff.view(np.int64).copy()).to(dev, non_blocking=True)
            d_p = torch.from_numpy(np.frombuffer(b"".join(pks[:n]), dtype=np.uint8).copy()).to(dev, non_blocking=True)
            d_s = torch.from_numpy(np.frombuffer(b"".join(sigs[:n]), dtype=np.uint8).copy()).to(dev, non_blocking=True)
            d_ok = torch.zeros(n, dtype=torch.uint8, device=dev)
            eng.verify_batch_dev(group, d_m.data_ptr(), d_o.data_ptr(), d_p.data_ptr(), d_s.data_ptr(), 0, d_ok.data_ptr(), n, stream=st.cuda_stream)
        st.synchronize()
        assert [bool(x) for x in d_ok.cpu().tolist()] == expect[:n], n


def test_fused_sign_batches_match_the_oracle(eng):
    """blsmi_g{2,1}pubs_sign_batch / sign_with_domain_batch: sig_i = sk_i * H(m_i) in ONE call (g2pubs/bls.go:132-135, g1pubs/bls.go:132-141) --
    the reference's signatures byte for byte on the small-call path and on the throughput kernels, ragged and empty messages, the zero key
    (signature at infinity, flagged), and every signature verifies under its PrivToPub key through the batch verify."""
    xs = P.XORShift(907)
    dom = bytes(range(8))
    for n in (1, 5, 700):
        sks = [sk_bytes(xs) for _ in range(n)]
        msgs = [(b"m%d" % i) * (i % 7) for i in range(n)]                      # lengths 0, 2, 4, ... (message 0 is empty)
        m32 = [bytes([i & 255, i >> 8]) * 16 for i in range(n)]
        if n > 1:
            sks[1] = bytes(32)                                                 # sk = 0: the point at infinity
        s2, i2 = eng.g2pubs_sign_batch(msgs, b"".join(sks))
        s1, i1 = eng.g1pubs_sign_batch(msgs, b"".join(sks))
        sd, idm = eng.g1pubs_sign_with_domain_batch(m32, dom, b"".join(sks))
        check = range(n) if n <= 5 else (0, 1, 2, 63, 64, 65, 333, n - 1)
        for i in check:
            zero = n > 1 and i == 1
            assert bool(i2[i]) == zero and bool(i1[i]) == zero and bool(idm[i]) == zero, (n, i)
            if zero:
                assert not s2[i].any() and not s1[i].any() and not sd[i].any()
                continue
            assert s2[i].tobytes() == RC.g2pubs.sign(msgs[i], sks[i]), (n, i)
            assert s1[i].tobytes() == RC.g1pubs.sign(msgs[i], sks[i]), (n, i)
            assert sd[i].tobytes() == RC.g1pubs.sign_with_domain(m32[i], sks[i], dom), (n, i)
        # every signature verifies under its own key (the whole batch; the zero key's tuple is rejected: infinity on both sides)
        pk2, _ = eng.g2_mul_generator_batch(b"".join(sks), n)
        ok, _ = eng.g2pubs_verify_batch(msgs, pk2.reshape(-1), s2.reshape(-1))
        want = [not (n > 1 and i == 1) for i in range(n)]
        assert [bool(x) for x in ok] == want
        pk1, _ = eng.g1_mul_generator_batch(b"".join(sks), n)
        ok = eng.g1pubs_verify_with_domain_batch(m32, dom, pk1.reshape(-1), sd.reshape(-1))
        assert [bool(x) for x in ok] == want
    # argument errors
    lib = eng._lib()
    assert lib.blsmi_g2pubs_sign_batch(None, None, None, None, None, 3) != 0
    assert lib.blsmi_g2pubs_sign_batch(None, None, None, None, None, 0) == 0


def test_large_g2pubs_aggregate_with_the_cofactor_moved_through_the_pairing(eng):
    """From 65 536 messages a g2pubs VerifyAggregate (g2pubs/bls.go:240-270) hashes WITHOUT the cofactor clearing of hash.go:306-309 and raises
    the product of its Miller values to 1 - x once (program powc12raw; tests/test_lat_program.py has the identity on the oracle).  Verdicts --
    valid, one wrong key, one wrong message, a tampered aggregate -- equal those of the cleared-hash path (blsmi_set_option("agg_cofactor_pow", 0); BLSMI_AGG_COFACTOR_POW=0 at start-up, read
    per call), from host buffers and from resident ones; below the threshold the program is not used."""
    import ctypes
    import hashlib
    import os
    import torch
    n = 65536 + 4096 + 7
    nk = 64
    sk = [hashlib.sha256(b"powc-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk)]
    pks, _ = eng.g2_mul_generator_batch(b"".join(sk), nk)
    msgs = [b"aggregate message %d" % i for i in range(n)]
    packed = eng.PackedMsgs(msgs)
    sks = (b"".join(sk) * (n // nk + 1))[:32 * n]
    sigs, _ = eng.g2pubs_sign_batch(packed, sks)
    assert sigs[12345].tobytes() == RC.g2pubs.sign(msgs[12345], sk[12345 % nk])
    agg = eng.g1_sum(sigs.reshape(-1), n)
    allpk = np.ascontiguousarray(np.tile(pks, (n // nk + 1, 1))[:n]).reshape(-1)
    bad_pk = allpk.copy(); j = 40000; bad_pk[192 * j:192 * (j + 1)] = pks[(j + 1) % nk]
    bad_msgs = list(msgs); bad_msgs[777] = b"another message"
    agg2 = eng.g1_sum(sigs[:n - 1].reshape(-1), n - 1)
    lib = eng._lib()

    def run(m, p, a):
        lib.blsmi_set_profiling(1)
        v = eng.g2pubs_verify_aggregate(m, p, a)
        buf = ctypes.create_string_buffer(8192); lib.blsmi_last_profile(buf, ctypes.c_size_t(8192)); lib.blsmi_set_profiling(0)
        return v, buf.value.decode()
    try:
        for mode in (1, 0):
            eng.set_option("agg_cofactor_pow", mode)
            v, prof = run(packed, allpk, agg)
            assert v is True and ("k_lat:powc12raw" in prof) == (mode == 1), (mode, prof)
            assert run(packed, bad_pk, agg)[0] is False
            assert run(bad_msgs, allpk, agg)[0] is False
            assert run(packed, allpk, agg2)[0] is False
            # resident inputs
            dev = torch.device("cuda", 0)
            d_m = torch.from_numpy(packed.buf.copy()).to(dev); d_o = torch.from_numpy(packed.off.view(np.int64).copy()).to(dev)
            d_p = torch.from_numpy(allpk).to(dev); d_b = torch.from_numpy(bad_pk).to(dev)
            assert eng.verify_aggregate_dev("g2pubs", d_m.data_ptr(), d_o.data_ptr(), d_p.data_ptr(), agg, n) is True
            assert eng.verify_aggregate_dev("g2pubs", d_m.data_ptr(), d_o.data_ptr(), d_b.data_ptr(), agg, n) is False
        # partial products across "ranks" (bls_amd/dist.py's exchange, in process): a shard of 65 536 messages takes the new path, the rest of
        # the messages the cleared one; their product finishes to the same verdict
        eng.set_option("agg_cofactor_pow", 1)
        g2gen = np.frombuffer(RC.g2_generator(), dtype=np.uint8)
        for keys, want in ((allpk, True), (bad_pk, False)):
            pa, ba = eng.aggregate_partial("g2pubs", msgs[:65536], keys[:192 * 65536])
            pb, bb = eng.aggregate_partial("g2pubs", msgs[65536:], keys[192 * 65536:])
            assert not ba and not bb
            rhs = eng.fq12_product(np.concatenate([np.asarray(pa, dtype=np.uint64).reshape(-1), np.asarray(pb, dtype=np.uint64).reshape(-1)]))
            lhs = eng.miller_loop_batch(agg, g2gen, 1)[0]
            fe = eng.final_exponentiation_batch(np.stack([lhs, np.asarray(rhs, dtype=np.uint64).reshape(-1)]))
            assert bool(np.array_equal(fe[0], fe[1])) is want
        m2 = 65535
        a3 = eng.g1_sum(sigs[:m2].reshape(-1), m2)
        v, prof = run(msgs[:m2], allpk[:192 * m2], a3)
        assert v is True and "k_lat:powc12raw" not in prof
    finally:
        eng.set_option("agg_cofactor_pow", 1)
