"""-m gpu, round 5: the rolled level program (K_REP loops in k_lat.hip), blsmi_set_option, the environment read once."""
import numpy as np
import pytest

from gpu_common import P, RC, rand_g1, rand_g2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from bls_amd import engine
    engine.init(0)
    yield engine
    engine.set_option("lat_rolled", 1)


def test_rolled_pairing_program_on_the_device(eng):
    """pairing1 (default) against pairing1s, its straight-line copy: the squaring runs of the five ExpByX as loops the kernel repeats (K_REP; the reference's own loop, fq12.go:112-118).  Same
    Fq12 bits as the straight-line program and as the oracle, at one tuple, a ragged handful and a batch that queues several waves per SIMD."""
    xs = P.XORShift(5201)
    base = 6
    w1 = [rand_g1(xs) for _ in range(base)]; w2 = [rand_g2(xs) for _ in range(base)]
    want = RC.pairing_batch(b"".join(w1), b"".join(w2), base)
    try:
        for n in (1, 5, 70, 3000):
            a = b"".join(w1[i % base] for i in range(n)); b = b"".join(w2[i % base] for i in range(n))
            eng.set_option("lat_rolled", 0)
            straight = eng.pairing_batch(a, b, n)
            eng.set_option("lat_rolled", 1)
            rolled = eng.pairing_batch(a, b, n)
            assert np.array_equal(rolled, straight)
            assert np.array_equal(rolled[:min(n, base)], want[:min(n, base)])
            assert np.array_equal(rolled[n - 1], want[(n - 1) % base])
    finally:
        eng.set_option("lat_rolled", 1)


def test_set_option_rejects_unknown_names(eng):
    lib = eng._lib()
    assert lib.blsmi_set_option(b"no_such_option", 1) == -3 and lib.blsmi_set_option(None, 1) == -3
    for name in ("agg_cofactor_pow", "msm_sort", "dup_force_sort", "lat_rolled", "crowd_quad", "combine_mid_max"):
        assert lib.blsmi_set_option(name.encode(), 1) == 0
    eng.set_option("dup_force_sort", 0); eng.set_option("combine_mid_max", 8192)


def test_environment_is_read_at_initialisation_only(eng):
    """BLSMI_* variables changed AFTER the library initialised change nothing (include/blsmi.h "Environment"): a large g2pubs aggregate keeps
    taking the cofactor-power program whatever os.environ says; blsmi_set_option is what switches it."""
    import ctypes
    import hashlib
    import os
    n = 65536
    nk = 64
    sk = b"".join(hashlib.sha256(b"env-sk-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
    pks, _ = eng.g2_mul_generator_batch(sk, nk)
    msgs = eng.PackedMsgs([hashlib.sha256(b"env %d" % i).digest() for i in range(n)])
    sigs, _ = eng.g2pubs_sign_batch(msgs, sk * (n // nk))
    agg = eng.g1_sum(sigs.reshape(-1), n)
    allpk = np.ascontiguousarray(np.tile(pks, (n // nk, 1))).reshape(-1)
    lib = eng._lib()

    def prof_of_call():
        lib.blsmi_set_profiling(1)
        v = eng.g2pubs_verify_aggregate(msgs, allpk, agg)
        buf = ctypes.create_string_buffer(8192); lib.blsmi_last_profile(buf, ctypes.c_size_t(8192)); lib.blsmi_set_profiling(0)
        return v, buf.value.decode()
    saved = os.environ.get("BLSMI_AGG_COFACTOR_POW")
    try:
        os.environ["BLSMI_AGG_COFACTOR_POW"] = "0"
        v, prof = prof_of_call()
        assert v is True and "k_lat:powc12raw" in prof                        # the environment is not consulted any more
        eng.set_option("agg_cofactor_pow", 0)
        v, prof = prof_of_call()
        assert v is True and "k_lat:powc12raw" not in prof
    finally:
        eng.set_option("agg_cofactor_pow", 1)
        if saved is None:
            os.environ.pop("BLSMI_AGG_COFACTOR_POW", None)
        else:
            os.environ["BLSMI_AGG_COFACTOR_POW"] = saved


def test_uncleared_hash_path_flags_cancelling_map_points(eng):
    """a large g2pubs VerifyAggregate pairs the hash points BEFORE their cofactor clearing (hash.cuh: swu_finish_g1 with clear = 0)
    and raises the Miller product to 1 - x.  That identity needs S on the curve; when a message's two SWU points cancel (p2 = -p1, ~2^-381) the
    reference's affine steps produce a value out of (0, 0) and the identity does not apply: the kernel raises a flag and the host redoes the call
    with cleared points.  Here the throughput tail runs on chosen mapped points: the flag comes up exactly for the cancelling pair, the other
    messages' uncleared points S satisfy [1 - x] S = the cleared point, and with clear = 1 nothing is flagged."""
    import ctypes
    lib = eng._lib()
    xs = P.XORShift(5210)
    # mapped points of real messages: the two SWU points per message, through the library's own first half
    msgs = [b"special %d" % i for i in range(6)]
    h = eng.hash_g1_batch(msgs)
    # recover usable (p1, p2) pairs on the 11-isogenous curve: any two points of E11 serve -- take them from the debug SWU op on random t
    from gpu_common import pack, rand_fq
    ts = rand_fq(xs, 12)
    rec = np.zeros((12, 18), dtype=np.uint64)
    for i, t in enumerate(ts):
        rec[i, :6] = pack([t])
    swu, _ = eng.debug_op("SWU_G1", rec.reshape(-1))
    pts = []
    for i in range(12):
        x = P.from_mont(P.from_limbs64(swu.reshape(12, 18)[i, :6])); y = P.from_mont(P.from_limbs64(swu.reshape(12, 18)[i, 6:12]))
        pts.append((x, y))
    def wire(p):
        return p[0].to_bytes(48, "big") + p[1].to_bytes(48, "big")
    pairs = [wire(pts[2 * i]) + wire(pts[2 * i + 1]) for i in range(6)]
    neg = (pts[4][0], (P.Q - pts[4][1]) % P.Q)
    pairs[2] = wire(pts[4]) + wire(neg)                                       # message 2: p2 = -p1
    buf = np.frombuffer(b"".join(pairs), dtype=np.uint8)

    def finish(clear):
        out = np.zeros(96 * 6, dtype=np.uint8); sp = ctypes.c_int(0)
        assert lib.blsmi_debug_hash_g1_finish(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), clear, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(sp), ctypes.c_size_t(6)) == 0
        return out.reshape(6, 96), sp.value
    cleared, sp1 = finish(1)
    raw, sp0 = finish(0)
    assert sp1 == 0 and sp0 == 2
    ok_pairs = [i for i in range(6) if i != 2]
    flagless = np.frombuffer(b"".join(pairs[i] for i in ok_pairs), dtype=np.uint8)
    out = np.zeros(96 * 5, dtype=np.uint8); sp = ctypes.c_int(0)
    assert lib.blsmi_debug_hash_g1_finish(flagless.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), 0, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.byref(sp), ctypes.c_size_t(5)) == 0
    assert sp.value == 0
    # [1 - x] S == the cleared point, for the messages the identity covers (1 - x = 1 + |x|; any-point ladder: S is outside the subgroup)
    k = (0xd201000000010000 + 1).to_bytes(32, "big")
    mul, inf = eng.g1_mul_batch(b"".join(raw[i].tobytes() for i in ok_pairs), k * 5, 5, any_point=True)
    assert not inf.any() and [m.tobytes() for m in mul] == [cleared[i].tobytes() for i in ok_pairs]
    assert h.shape[0] == 6


@pytest.mark.parametrize("group", ["g2pubs", "g1pubs"])
def test_layout_follows_the_load_of_the_device(eng, group):
    """blsmi 0.6: the layout of a mid-size call goes by what its DEVICE carries (blsmi.hip: call_load / use_quad; tools/midsize_concurrency.py) --
    a call of >= crowd_floor tuples takes the lane-quad kernels when other calls' tuples are in flight; alone it takes the lane-row kernels (round 6; the one-tuple-per-wave path below 2 048 tuples).
    Same verdicts (and the same Fq12 bits for Pairing) on either; "assume_load" stands in for the other callers; calls below the floor never move."""
    import ctypes
    from test_gpu_verify import _tuples
    msgs, pks, sigs, expect = _tuples(group, 96, 91)
    o = RC.g2pubs if group == "g2pubs" else RC.g1pubs
    for i in (0, 5, 9):
        assert o.verify(msgs[i], pks[i], sigs[i]) == expect[i]
    n = 2600
    sel = [i % 96 for i in range(n)]
    M = eng.PackedMsgs([msgs[i] for i in sel]); A = b"".join(pks[i] for i in sel); B = b"".join(sigs[i] for i in sel)
    want = [expect[i] for i in sel]
    fn = eng.g2pubs_verify_batch if group == "g2pubs" else eng.g1pubs_verify_batch
    lib = eng._lib()

    def call(f):
        lib.blsmi_set_profiling(1)
        r = f()
        buf = ctypes.create_string_buffer(8192); lib.blsmi_last_profile(buf, ctypes.c_size_t(8192)); lib.blsmi_set_profiling(0)
        return r, buf.value.decode()

    def quad(prof):                                                            # the PAIRING kernels' layout (the hash kernels have quad / row forms of their own: k_hash_g1_finish_quad, k_swu_g1_rows)
        return "k_miller2_quad" in prof or "k_miller1h_quad" in prof or "k_final_exp_is_one_quad" in prof

    def row(prof):
        return "k_miller2_row" in prof or "k_miller1m_row" in prof or "k_miller1s_row" in prof or "k_final_exp_is_one_row" in prof
    try:
        (ok, _), prof = call(lambda: fn(M, A, B))
        assert list(ok) == want and "k_miller1m_row" in prof and not quad(prof), prof      # alone: the lane-row layout since round 6 (2 304 .. 8 192 tuples; the wave path below)
        eng.set_option("assume_load", 4000)
        (ok, _), prof = call(lambda: fn(M, A, B))
        assert list(ok) == want and quad(prof) and not row(prof) and "k_lat:verify" not in prof and "k_lat:hashfin" not in prof, prof
        eng.set_option("crowd_quad", 0)
        (ok, _), prof = call(lambda: fn(M, A, B))
        assert list(ok) == want and not quad(prof), prof
        eng.set_option("crowd_quad", 1)
        m = 1000                                                                # below the floor: a small call keeps its latency whatever else runs
        (ok, _), prof = call(lambda: fn(eng.PackedMsgs([msgs[i] for i in sel[:m]]), A[:len(pks[0]) * m], B[:len(sigs[0]) * m]))
        assert list(ok) == want[:m] and not quad(prof), prof
        if group == "g2pubs":                                                   # Pairing: the same Fq12 bits from both layouts
            g1 = np.frombuffer(B, dtype=np.uint8); g2 = np.frombuffer(A, dtype=np.uint8)
            crowded, prof = call(lambda: eng.pairing_batch(g1, g2, n))
            assert "k_miller1h_quad" in prof, prof
            eng.set_option("assume_load", 0)
            alone, prof = call(lambda: eng.pairing_batch(g1, g2, n))
            assert "k_miller1h_row" in prof and np.array_equal(crowded, alone), prof
            eng.set_row_threshold(0, 0)                                         # without the row layout: the wave path, as before round 6
            wave, prof = call(lambda: eng.pairing_batch(g1, g2, n))
            assert "k_lat:pairing1" in prof and np.array_equal(wave, alone), prof
            eng.set_row_threshold(*eng.ROW_DEFAULT)
            assert alone[7].tobytes() == RC.pairing_batch(g1[96 * 7:96 * 8].tobytes(), g2[192 * 7:192 * 8].tobytes(), 1).tobytes()
    finally:
        eng.set_option("assume_load", 0); eng.set_option("crowd_quad", 1); eng.set_row_threshold(*eng.ROW_DEFAULT)
    # concurrent callers: the load is taken off the device when a call ends (nothing left behind for the next lone caller)
    import threading
    errs = []

    def worker():
        try:
            for _ in range(3):
                ok, _ = fn(M, A, B)
                assert list(ok) == want
        except Exception as e:                                                  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=worker) for _ in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    (ok, _), prof = call(lambda: fn(M, A, B))
    assert list(ok) == want and not quad(prof), prof


def test_concurrent_mid_size_verify_calls_merge_and_keep_their_verdicts(eng):
    """verify_host.inc, the combiner's mid-size class: concurrent Verify calls of 1 024 ... 8 191 tuples are merged into one launch among themselves
    (one-tuple calls keep their own queue).  Every caller gets exactly its own verdicts, in both point formats, whatever it was merged with;
    blsmi_set_option("combine_mid_max", 0) switches the class off and changes nothing in the results."""
    import threading
    from gpu_common import g1_to_jac, g2_to_jac
    from test_gpu_verify import _tuples
    msgs, pks, sigs, expect = _tuples("g2pubs", 64, 123)
    jp = [g2_to_jac(p, (3 + i, 2 * i + 1)) for i, p in enumerate(pks)]
    js = [g1_to_jac(s, 11 + i) for i, s in enumerate(sigs)]
    errs, done = [], []

    def caller(seed, n, jac, rounds):
        try:
            rng = np.random.default_rng(seed)
            for _ in range(rounds):
                sel = rng.integers(64, size=n)
                M = eng.PackedMsgs([msgs[i] for i in sel])
                if jac:
                    ok, _ = eng.g2pubs_verify_batch_jac(M, b"".join(jp[i] for i in sel), b"".join(js[i] for i in sel))
                else:
                    ok, _ = eng.g2pubs_verify_batch(M, b"".join(pks[i] for i in sel), b"".join(sigs[i] for i in sel))
                assert list(ok) == [expect[i] for i in sel], (seed, n, jac)
            done.append(seed)
        except Exception as e:                                                  # noqa: BLE001
            errs.append(e)
    assert any(expect) and not all(expect)
    for mid in (8192, 0):
        eng.set_option("combine_mid_max", mid)
        try:
            shapes = [(1, 1500, True, 4), (2, 3000, True, 4), (3, 5000, True, 3), (4, 1, True, 12), (5, 2600, False, 4), (6, 1100, False, 4), (7, 7000, True, 2), (8, 3, False, 12)]
            ts = [threading.Thread(target=caller, args=a) for a in shapes]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            assert not errs, errs
        finally:
            eng.set_option("combine_mid_max", 8192)
    assert len(done) == 16
