"""-m gpu: the lane-ROW layout (sixteen lanes per tuple, bls_amd/csrc/row_body.inc, k_pairing_row.hip) -- the layout for the call sizes the
reference's API produces: a few thousand tuples (4 096 tuples = one wave per SIMD).  Every Fq12 routine op by op
against the oracle, then Pairing and Verify on all FOUR paths (one tuple per wave / per lane row / per lane quad / per lane pair):
bit-identical Fq12 (pairing.go:132-136, fq12.go:27-237), identical verdicts (g2pubs/bls.go:159-162, g1pubs/bls.go:165-168)."""
import numpy as np
import pytest

from gpu_common import P, RC, pack, rand_fq, rand_g1, rand_g2, sk_bytes

pytestmark = pytest.mark.gpu
OCT_MAX_DEFAULT = 7168                                                          # blsmi.hip: g_hash_oct_max


@pytest.fixture(scope="module")
def eng():
    from bls_amd import engine
    engine.init(0)
    yield engine
    engine.set_latency_threshold(8192); engine.set_quad_threshold(16384); engine.set_row_threshold(*engine.ROW_DEFAULT)


def _rand_rec(xs, n, width):
    return np.stack([pack(rand_fq(xs, width)) for _ in range(n)])


def _cyclotomic(recs):
    """x^((q^6-1)(q^2+1)) of every record, from the oracle: elements of the cyclotomic subgroup"""
    cyc = []
    for x in recs:
        inv = RC.fq12_inverse(x)[1]
        conj = x.copy().reshape(12, 6)
        for k in range(6, 12):
            conj[k] = RC.fq_neg(conj[k])
        t = RC.fq12_mul(conj.reshape(-1), inv)
        cyc.append(RC.fq12_mul(RC.fq12_frobenius(t, 2), t))
    return np.stack(cyc)


def test_fq12_ops_in_lane_row_layout(eng):
    """n = 11: a ragged final workgroup (4 tuples per workgroup); values with coefficient patterns that exercise every pair of a row"""
    xs = P.XORShift(6101)
    n = 11
    a12 = _rand_rec(xs, n, 12); b12 = _rand_rec(xs, n, 12)
    # sparse operands: one non-zero Fq2 coefficient each (w^p alone), so that a wrong route between pairs cannot cancel out
    for k in range(6):
        a12[k] = 0; a12[k][12 * k:12 * k + 12] = pack(rand_fq(xs, 2))
    for name, ref in [("FQ12_MUL", lambda x, y: RC.fq12_mul(x, y)), ("FQ12_SQR", lambda x, y: RC.fq12_sqr(x)), ("FQ12_INV", lambda x, y: RC.fq12_inverse(x)[1]),
                      ("FQ12_FROB1", lambda x, y: RC.fq12_frobenius(x, 1)), ("FQ12_FROB2", lambda x, y: RC.fq12_frobenius(x, 2)), ("FQ12_FROB3", lambda x, y: RC.fq12_frobenius(x, 3))]:
        out, _ = eng.debug_op(name, a12, b12 if name == "FQ12_MUL" else None, lane_row=True)
        want = np.stack([ref(x, y) for x, y in zip(a12, b12)])
        bad = [i for i in range(n) if not np.array_equal(out[i], want[i])]
        assert not bad, (name, bad)
    out, _ = eng.debug_op("FQ12_MUL", b12, a12, lane_row=True)             # the sparse operand on the other side
    assert np.array_equal(out, np.stack([RC.fq12_mul(x, y) for x, y in zip(b12, a12)]))
    # the sparse line multiplication (fq12.go:32-47): (c0, c1, c4) = b[0..5]
    out, _ = eng.debug_op("FQ12_MUL_BY_014", a12, b12, lane_row=True)
    want = np.stack([RC.fq12_mul_by_014(x, y[0:12], y[12:24], y[24:36]) for x, y in zip(a12, b12)])
    assert np.array_equal(out, want)
    cyc = _cyclotomic(b12[:9])
    out, _ = eng.debug_op("FQ12_CYCLO_SQR", cyc, lane_row=True)
    assert np.array_equal(out, np.stack([RC.fq12_sqr(x) for x in cyc]))
    one = np.zeros((3, 72), dtype=np.uint64); one[:, :6] = pack([1])
    for name in ("FQ12_CYCLO_SQR", "FQ12_SQR", "FQ12_INV", "FQ12_FROB1"):
        out, _ = eng.debug_op(name, one, lane_row=True)
        assert np.array_equal(out, one), name


def test_miller_loop_steps_in_lane_row_layout(eng):
    """one doubling step and one mixed addition step of the homogeneous Miller loop (pairing_body.inc: doubling_step_h_i, addition_step_h) on random
    projective points: the row form (two / four product times over the eight pairs) against the lane-pair routine run by every pair alike --
    same new point, same line at P, canonical bits"""
    xs = P.XORShift(6105)
    recs = _rand_rec(xs, 9, 12)                                            # (X, Y, Z, xq, yq, xP | yP): any field elements do, the formulas are polynomial
    for name in ("ROW_DBL_STEP", "ROW_ADD_STEP"):
        got, _ = eng.debug_op(name, recs, lane_row=True)
        ref, _ = eng.debug_op(name + "_REF", recs, lane_row=True)
        bad = [what for k, what in enumerate(("X3", "Y3", "Z3", "c0", "c1", "c4")) if not np.array_equal(got[:, 12 * k:12 * k + 12], ref[:, 12 * k:12 * k + 12])]
        assert not bad, (name, bad)


def test_pairing_on_the_four_paths_agrees_with_the_oracle(eng):
    """the same tuples through the lane-row kernels at ragged sizes around their 4-tuple workgroups, and through the other three layouts:
    the reference's generator vector (pairing_test.go:9-58) and points outside the subgroup included"""
    from test_gpu_round3 import _torsion_points
    xs = P.XORShift(6102)
    g1s, g2s = _torsion_points()
    a = [RC.g1_generator()] + [rand_g1(xs) for _ in range(20)] + g1s[:3]
    b = [RC.g2_generator()] + [rand_g2(xs) for _ in range(20)] + g2s[:3]
    n = len(a)
    want = RC.pairing_batch(b"".join(a), b"".join(b), n)
    try:
        eng.set_row_threshold(1, 1 << 20)                                          # lane row, whatever the size
        for m in (1, 3, 4, 5, n):
            got = eng.pairing_batch(b"".join(a[:m]), b"".join(b[:m]), m)
            assert np.array_equal(got, want[:m]), ("row", m)
        eng.set_row_threshold(0, 0)
        eng.set_latency_threshold(0); eng.set_quad_threshold(1 << 20)              # lane quad
        assert np.array_equal(eng.pairing_batch(b"".join(a), b"".join(b), n), want)
        eng.set_quad_threshold(0)                                                  # lane pair
        assert np.array_equal(eng.pairing_batch(b"".join(a), b"".join(b), n), want)
        eng.set_latency_threshold(8192)                                            # one tuple per wave
        assert np.array_equal(eng.pairing_batch(b"".join(a), b"".join(b), n), want)
    finally:
        eng.set_latency_threshold(8192); eng.set_quad_threshold(16384); eng.set_row_threshold(*eng.ROW_DEFAULT)


def _g2pubs_tuples(n, seed, every):
    xs = P.XORShift(seed)
    msgs, pks, sigs, expect = [], [], [], []
    for i in range(n):
        sk = sk_bytes(xs)
        m = b"row layout %d" % i
        pk, sig = RC.g2pubs.priv_to_pub(sk), RC.g2pubs.sign(m, sk)
        good = True
        if i % every == every - 1:
            good = False
            kind = (i // every) % 3
            if kind == 0:
                m = m + b"!"
            elif kind == 1:
                pk = RC.g2pubs.priv_to_pub(sk_bytes(xs))
            else:
                sig = sig[:48] + ((P.Q - int.from_bytes(sig[48:], "big")) % P.Q).to_bytes(48, "big")
        msgs.append(m); pks.append(pk); sigs.append(sig); expect.append(good)
    return msgs, pks, sigs, expect


def _g1pubs_tuples(n, seed, every):
    xs = P.XORShift(seed)
    msgs, pks, sigs, expect = [], [], [], []
    for i in range(n):
        sk = sk_bytes(xs)
        m = b"row layout g1pubs %d" % i
        pk, sig = RC.g1pubs.priv_to_pub(sk), RC.g1pubs.sign(m, sk)
        good = True
        if i % every == every - 1:
            good = False
            if (i // every) % 2 == 0:
                m = m + b"?"
            else:
                pk = RC.g1pubs.priv_to_pub(sk_bytes(xs))
        msgs.append(m); pks.append(pk); sigs.append(sig); expect.append(good)
    return msgs, pks, sigs, expect


@pytest.mark.parametrize("side", [1, 0])
@pytest.mark.parametrize("group", ["g2pubs", "g1pubs"])
def test_verify_in_the_row_layout(eng, group, side):
    """Verify of both packages forced into the row kernels: the oracle's verdict table, ragged sizes.  side = 1 (the default): the signature side's
    Miller loop on a side stream beside the hash (k_miller1s_row; g2pubs: over the generator's prepared lines), then k_miller1m_row times that value;
    side = 0: one two-pair loop (k_miller2_row).  k_final_exp_is_one_row either way"""
    msgs, pks, sigs, expect = (_g2pubs_tuples if group == "g2pubs" else _g1pubs_tuples)(13, 6103, 3)
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    assert [o.verify(m, p, s) for m, p, s in zip(msgs, pks, sigs)] == expect
    fn = eng.g2pubs_verify_batch if group == "g2pubs" else eng.g1pubs_verify_batch
    try:
        eng.set_row_threshold(1, 1 << 20); eng.set_option("row_side", side)
        for m in (1, 4, 5, 13):
            ok, _ = fn(msgs[:m], b"".join(pks[:m]), b"".join(sigs[:m]))
            assert list(ok) == expect[:m], (group, m, side)
        eng.set_option("row_side_piece", 4)                                  # the side kernel in pieces: 4 + 4 + 4 + 1 tuples
        ok, _ = fn(msgs, b"".join(pks), b"".join(sigs))
        assert list(ok) == expect, (group, side, "pieces")
    finally:
        eng.set_row_threshold(*eng.ROW_DEFAULT); eng.set_option("row_side", 1); eng.set_option("row_side_piece", 0)


def test_row_layout_at_its_design_size(eng):
    """4 096 pairings (1 024 waves: one per SIMD) on the DEFAULT thresholds take the row kernels: every record against the lane-pair
    kernels' output, a spread sample against the oracle; 4 096 g2pubs verifies with a corruption schedule likewise"""
    n = 4096
    base = 128
    xs = P.XORShift(6104)
    ka = b"".join(sk_bytes(xs) for _ in range(base)); kb = b"".join(sk_bytes(xs) for _ in range(base))
    g1b, _ = eng.g1_mul_generator_batch(ka, base); g2b, _ = eng.g2_mul_generator_batch(kb, base)
    reps = n // base
    g1 = np.ascontiguousarray(np.tile(g1b, (reps, 1)))
    g2 = np.ascontiguousarray(np.concatenate([np.roll(g2b, -r, axis=0) for r in range(reps)]))
    lib = __import__("bls_amd._native", fromlist=["load"]).load()
    import bench
    lib.blsmi_set_profiling(1); bench.read_profile(lib)
    got = eng.pairing_batch(g1.reshape(-1), g2.reshape(-1), n)
    lib.blsmi_set_profiling(0)
    prof = bench.read_profile(lib)
    assert "k_miller1h_row" in prof and "k_final_exp_row" in prof, prof
    try:
        eng.set_row_threshold(0, 0); eng.set_latency_threshold(0); eng.set_quad_threshold(0)
        ref = eng.pairing_batch(g1.reshape(-1), g2.reshape(-1), n)
    finally:
        eng.set_latency_threshold(8192); eng.set_quad_threshold(16384); eng.set_row_threshold(*eng.ROW_DEFAULT)
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert bad.size == 0, bad[:8]
    for i in (0, 1, 3, 4, 1025, n - 5, n - 1):
        assert np.array_equal(got[i], RC.pairing_batch(g1[i].tobytes(), g2[i].tobytes(), 1)[0]), i
    # verify: 128 signers x 32 messages each, every 7th tuple carries the wrong key
    sks = [ka[32 * i:32 * i + 32] for i in range(base)]
    pks, _ = eng.g2_mul_generator_batch(ka, base)
    msgs = [b"row design size %d" % i for i in range(n)]
    h = eng.hash_g1_batch(msgs)
    sigs, _ = eng.g1_mul_batch(h.reshape(-1), b"".join(sks[i % base] for i in range(n)), n)
    allpk = np.stack([pks[(i + (1 if i % 7 == 6 else 0)) % base] for i in range(n)])
    expect = [i % 7 != 6 for i in range(n)]
    lib.blsmi_set_profiling(1); bench.read_profile(lib)
    ok, _ = eng.g2pubs_verify_batch(msgs, allpk.reshape(-1), sigs.reshape(-1))
    lib.blsmi_set_profiling(0)
    prof = bench.read_profile(lib)
    assert "k_miller1m_row" in prof and "k_final_exp_is_one_row" in prof, prof     # (the signature side beside the hash: verify_host.inc, verify_sig_side_start)
    assert list(ok) == expect
    try:                                                                    # the side kernel in ragged pieces (as a call of more than 4 096 tuples runs it), and the two-pair loop
        eng.set_option("row_side_piece", 1000)
        ok2, _ = eng.g2pubs_verify_batch(msgs, allpk.reshape(-1), sigs.reshape(-1))
        eng.set_option("row_side_g2pubs", 0)
        lib.blsmi_set_profiling(1); bench.read_profile(lib)
        ok3, _ = eng.g2pubs_verify_batch(msgs, allpk.reshape(-1), sigs.reshape(-1))
        lib.blsmi_set_profiling(0)
        assert "k_miller2_row" in bench.read_profile(lib)
    finally:
        eng.set_option("row_side_piece", 0); eng.set_option("row_side_g2pubs", 1)
    assert list(ok2) == expect and list(ok3) == expect
    for i in (0, 6, 4095):
        assert RC.g2pubs.verify(msgs[i], allpk[i].tobytes(), sigs[i].tobytes()) == expect[i]


@pytest.mark.parametrize("group", ["g2pubs", "g1pubs"])
def test_verify_aggregate_miller_loops_in_the_row_layout(eng, group):
    """VerifyAggregate over distinct messages (g2pubs/bls.go:240-270, g1pubs/bls.go:252-282) with its n Miller loops on the lane-row kernel
    (k_miller1s_row, one loop per row; product tree and final exponentiation as before): the oracle's verdicts at small n with the layout forced,
    and at 3 000 signers on the default thresholds -- true, false with one key replaced, equal to the one-tuple-per-wave path's verdicts"""
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    xs = P.XORShift(6106)
    n = 9
    sks = [sk_bytes(xs) for _ in range(n)]
    msgs = [b"row aggregate %d" % i for i in range(n)]
    pks = [o.priv_to_pub(k) for k in sks]; sigs = [o.sign(m, k) for m, k in zip(msgs, sks)]
    summ = eng.g1_sum if group == "g2pubs" else eng.g2_sum
    agg = summ(b"".join(sigs), n)
    fn = eng.g2pubs_verify_aggregate if group == "g2pubs" else eng.g1pubs_verify_aggregate
    lib = eng._lib()
    import bench
    try:
        eng.set_row_threshold(1, 1 << 20)
        assert o.verify_aggregate(agg, pks, msgs) is True
        lib.blsmi_set_profiling(1); bench.read_profile(lib)
        assert fn(msgs, b"".join(pks), agg) is True
        lib.blsmi_set_profiling(0)
        assert "k_miller1s_row" in bench.read_profile(lib)
        for m in (1, 4, 5):
            part = summ(b"".join(sigs[:m]), m)
            assert fn(msgs[:m], b"".join(pks[:m]), part) is True and o.verify_aggregate(part, pks[:m], msgs[:m]) is True
        wrong = list(pks); wrong[6] = pks[2]
        assert fn(msgs, b"".join(wrong), agg) is False and o.verify_aggregate(agg, wrong, msgs) is False
    finally:
        eng.set_row_threshold(*eng.ROW_DEFAULT)
    # 3 000 signers, default thresholds: the row kernel serves; the wave path (row layout off) must agree
    n = 3000
    nk = 64
    skb = b"".join(sk_bytes(xs) for _ in range(nk))
    msgs = [b"row aggregate big %d" % i for i in range(n)]
    if group == "g2pubs":
        pk, _ = eng.g2_mul_generator_batch(skb, nk); h = eng.hash_g1_batch(msgs); sg, _ = eng.g1_mul_batch(h.reshape(-1), (skb * (n // nk + 1))[:32 * n], n)
    else:
        pk, _ = eng.g1_mul_generator_batch(skb, nk); h = eng.hash_g2_batch(msgs); sg, _ = eng.g2_mul_batch(h.reshape(-1), (skb * (n // nk + 1))[:32 * n], n)
    allpk = np.ascontiguousarray(np.stack([pk[i % nk] for i in range(n)]))
    agg = summ(sg.reshape(-1), n)
    bad = allpk.copy(); bad[1234] = pk[(1234 + 1) % nk]
    try:
        lib.blsmi_set_profiling(1); bench.read_profile(lib)
        t = fn(msgs, allpk.reshape(-1), agg)
        lib.blsmi_set_profiling(0)
        assert "k_miller1s_row" in bench.read_profile(lib)
        f = fn(msgs, bad.reshape(-1), agg)
        eng.set_row_threshold(0, 0)
        assert (t, f) == (True, False) and fn(msgs, allpk.reshape(-1), agg) is True and fn(msgs, bad.reshape(-1), agg) is False
    finally:
        eng.set_row_threshold(*eng.ROW_DEFAULT)


def test_hash_g2_cofactor_clearing_in_the_row_layout(eng):
    """HashG2 (hash.go:391-411) of a few thousand messages: the maps and the 3-isogeny a lane pair per message (k_hash_g2_front), clearH2
    (hash.go:368-389) sixteen lanes per message (k_clear_h2_row, row_g2.inc).  Same 192 bytes as the lane-pair kernel for every message --
    ragged messages, a count that is not a multiple of the 4-message workgroups --, samples against the oracle; the row Jacobian formulas are
    the reference's (g2.go:389-529) without their special cases: a message that meets one is flagged and redone by the one-lane routine"""
    n = 2051
    msgs = [(b"row hash %d" % i) * (1 + i % 3) for i in range(n)]
    lib = eng._lib()
    import bench
    try:
        eng.set_option("hash_oct_max", 0)                                    # (the eight-lane tail takes precedence where its range covers the count: off for the first two)
        eng.set_option("hash_row_min", 1)
        lib.blsmi_set_profiling(1); bench.read_profile(lib)
        a = eng.hash_g2_batch(msgs)
        lib.blsmi_set_profiling(0)
        assert "k_clear_h2_row" in bench.read_profile(lib)
        small = eng.hash_g2_batch(msgs[:5])                                  # a ragged workgroup, 1 .. 4 messages a wave
        one = eng.hash_g2_batch(msgs[:1])
        eng.set_option("hash_row_max", 0); eng.set_option("hash_quad_min", 1)   # four lanes per message (k_clear_h2_quad, quad_g2.inc)
        lib.blsmi_set_profiling(1); bench.read_profile(lib)
        q = eng.hash_g2_batch(msgs)
        lib.blsmi_set_profiling(0)
        assert "k_clear_h2_quad" in bench.read_profile(lib)
        qsmall = eng.hash_g2_batch(msgs[:17])
        eng.set_option("hash_quad_max", 0)
        eng.set_option("hash_oct_min", 1); eng.set_option("hash_oct_max", 1 << 20)   # eight lanes per message (k_clear_h2_oct, oct_g2.inc), ragged around the eight-message workgroups
        lib.blsmi_set_profiling(1); bench.read_profile(lib)
        oc = eng.hash_g2_batch(msgs)
        lib.blsmi_set_profiling(0)
        assert "k_clear_h2_oct" in bench.read_profile(lib)
        osmall = [eng.hash_g2_batch(msgs[:k]) for k in (1, 7, 9, 17)]
        eng.set_option("hash_oct_max", 0)
        b = eng.hash_g2_batch(msgs)
    finally:
        eng.set_option("hash_oct_min", 2048); eng.set_option("hash_oct_max", OCT_MAX_DEFAULT)
        eng.set_option("hash_row_min", 2048); eng.set_option("hash_row_max", 4096); eng.set_option("hash_quad_min", 4097); eng.set_option("hash_quad_max", 16384)
    bad = np.nonzero((a != b).any(axis=1))[0]
    assert bad.size == 0, bad[:8]
    bad = np.nonzero((q != b).any(axis=1))[0]
    assert bad.size == 0, ("quad", bad[:8])
    assert np.array_equal(qsmall, b[:17])
    bad = np.nonzero((oc != b).any(axis=1))[0]
    assert bad.size == 0, ("oct", bad[:8])
    for k, o in zip((1, 7, 9, 17), osmall):
        assert np.array_equal(o, b[:k]), ("oct", k)
    assert np.array_equal(small, a[:5]) and np.array_equal(one, a[:1])
    for i in (0, 1, 3, 4, 1024, n - 2, n - 1):
        assert a[i].tobytes() == RC.hash_g2(msgs[i]), i


@pytest.mark.parametrize("layout", ["row", "quad"])
def test_row_g2_jacobian_arithmetic_against_the_oracle(eng, layout):
    """row_g2.inc (sixteen lanes per point) and quad_g2.inc (four: BLSMI_OP_LANE_QUAD with the same ops) one operation at a time (blsmi_debug_op, BLSMI_OP_ROW_G2_*): the doubling and the general addition on arbitrary field elements (the
    formulas are polynomial) against the oracle's g2.go:389-443 / 446-529, bit for bit in Jacobian coordinates; clearH2 of points on E' OUTSIDE the subgroup
    (what the isogeny hands it) against the pure-Python oracle's hash.go:368-389; and the exceptions the row formulas do not special-case -- infinity in,
    an addition of a point to itself or to infinity -- leave Z3 = 0, which is what k_clear_h2_row tests before handing the message to the one-lane routine"""
    from gpu_common import g2_to_jac, rand_z2
    from test_gpu_round3 import _torsion_points
    lay = dict(lane_row=True) if layout == "row" else dict(lane_quad=True)
    xs = P.XORShift(6107)
    recs = _rand_rec(xs, 9, 12)
    got, _ = eng.debug_op("ROW_G2_DOUBLE", recs, **lay)
    for i in range(9):
        assert np.array_equal(got[i][:36], RC.g2_double(recs[i][:36])), i
        assert not got[i][36:].any()
    got, _ = eng.debug_op("ROW_G2_ADD", recs, **lay)
    for i in range(9):
        assert np.array_equal(got[i][:36], RC.g2_add(recs[i][:36], recs[i][36:])), i
    # the exceptions: (P, P), (P, infinity), (infinity, P) -> Z3 = 0
    _, g2s = _torsion_points()
    pts = g2s[:3] + [rand_g2(xs) for _ in range(2)]
    jp = [np.frombuffer(g2_to_jac(w, rand_z2(xs)), dtype=np.uint64) for w in pts]
    inf = np.concatenate([jp[0][:24], np.zeros(12, np.uint64)])
    exc = np.stack([np.concatenate([jp[0], jp[0]]), np.concatenate([jp[1], inf]), np.concatenate([inf, jp[2]])])
    got, _ = eng.debug_op("ROW_G2_ADD", exc, **lay)
    assert not got[:, 24:36].any()
    # clearH2: the affine image of the result is the oracle's; Z = 0 in, Z = 0 out
    cl = np.stack([np.concatenate([j, j]) for j in jp] + [np.concatenate([inf, inf])])
    got, _ = eng.debug_op("ROW_CLEAR_H2", cl, **lay)
    f2 = lambda w: tuple((int.from_bytes(w[96 * k:96 * k + 48], "big"), int.from_bytes(w[96 * k + 48:96 * k + 96], "big")) for k in range(2))
    for i, w in enumerate(pts):
        assert f2(RC.g2_jac_to_affine_bytes(got[i][:36])) == P.clear_h2(f2(w)), i
    assert not got[len(pts)][24:36].any()


def test_hash_g1_tail_four_lanes_per_message(eng):
    """HashG1 (hash.go:306-331) of a few thousand messages: the two maps on two lanes (k_swu_g1_two_lanes), then the sum, the 11-isogeny and the cofactor
    clearing FOUR lanes per message (k_hash_g1_finish_quad, quad_g1.inc).  Same 96 bytes as the one-lane tail for every message of a ragged batch, samples
    against the oracle; on chosen mapped points the exceptions the quad formulas do not special-case -- two points that cancel, two that coincide -- are
    flagged and redone by the one-lane routine with the reference's special cases (g1.go:400-482)"""
    import ctypes
    n = 2051
    msgs = [(b"quad g1 %d" % i) * (1 + i % 3) for i in range(n)]
    lib = eng._lib()
    import bench
    try:
        eng.set_option("hash_g1_quad_min", 1)
        lib.blsmi_set_profiling(1); bench.read_profile(lib)
        a = eng.hash_g1_batch(msgs)
        lib.blsmi_set_profiling(0)
        assert "k_hash_g1_finish_quad" in bench.read_profile(lib)
        small = eng.hash_g1_batch(msgs[:17])
        eng.set_option("hash_g1_quad_max", 0)
        b = eng.hash_g1_batch(msgs)
    finally:
        eng.set_option("hash_g1_quad_min", 1280); eng.set_option("hash_g1_quad_max", 32768)
    bad = np.nonzero((a != b).any(axis=1))[0]
    assert bad.size == 0, bad[:8]
    assert np.array_equal(small, b[:17])
    for i in (0, 1, 3, 4, 1024, n - 2, n - 1):
        assert a[i].tobytes() == RC.hash_g1(msgs[i]), i
    # chosen mapped points (on the 11-isogenous curve: the SWU unit op on random t)
    xs = P.XORShift(6108)
    from gpu_common import pack, rand_fq
    ts = rand_fq(xs, 12)
    rec = np.zeros((12, 18), dtype=np.uint64)
    for i, t in enumerate(ts):
        rec[i, :6] = pack([t])
    swu, _ = eng.debug_op("SWU_G1", rec.reshape(-1))
    pts = [(P.from_mont(P.from_limbs64(swu.reshape(12, 18)[i, :6])), P.from_mont(P.from_limbs64(swu.reshape(12, 18)[i, 6:12]))) for i in range(12)]
    wire = lambda p: p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")
    pairs = [wire(pts[2 * i]) + wire(pts[2 * i + 1]) for i in range(6)]
    pairs[2] = wire(pts[4]) + wire((pts[4][0], (P.Q - pts[4][1]) % P.Q))      # message 2: p2 = -p1
    pairs[4] = wire(pts[8]) + wire(pts[8])                                    # message 4: p2 = p1
    buf = np.frombuffer(b"".join(pairs), dtype=np.uint8)

    def finish(clear):
        out = np.zeros(96 * 6, dtype=np.uint8); sp = ctypes.c_int(0)
        assert lib.blsmi_debug_hash_g1_finish(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), clear, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(sp), ctypes.c_size_t(6)) == 0
        return out.reshape(6, 96), sp.value
    want, _ = finish(1)
    got, redone = finish(2)
    assert redone == 2 and np.array_equal(got, want)
