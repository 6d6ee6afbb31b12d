"""Prepared public keys (blsmi 0.4, k_prepared_pair.hip): G2AffineToPrepared (g2.go:639-801) once into device memory, then Miller
loops that read a key's lines.  Parity: (i) the tables ARE the reference's G2Prepared.coeffs -- exported and compared limb for limb
with the oracle's restatement of G2AffineToPrepared; (ii) every prepared entry point returns what the unprepared one returns on the
same keys (pairing values bit for bit, verdict bytes, aggregate verdicts), which in turn is pinned to the oracle; both with small
batches on the latency programs (keys gathered back out of the tables) and on the throughput kernels that read the tables."""
import hashlib

import numpy as np
import pytest

from gpu_common import P, RC, rand_g1, rand_g2, sk_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["latency-path", "lane-row", "lane-quad", "lane-pair"])
def eng(request):
    from bls_amd import engine
    engine.init(0)
    # four paths, same results: one tuple per wave (k_lat.hip) / per lane quad (k_pairing_quad.hip) / per lane pair
    engine.set_latency_threshold(8192 if request.param in ("latency-path", "lane-row") else 0)
    engine.set_quad_threshold(0 if request.param == "lane-pair" else 16384)
    engine.set_row_threshold(*((1, 1 << 20) if request.param == "lane-row" else (0, 0)))   # round 6: sixteen lanes per tuple (k_pairing_row.hip), whatever the size
    yield engine
    engine.set_latency_threshold(8192); engine.set_quad_threshold(16384); engine.set_row_threshold(*engine.ROW_DEFAULT)


def _dev(x):
    import torch
    a = np.frombuffer(x, dtype=np.uint8).copy() if isinstance(x, (bytes, bytearray)) else np.ascontiguousarray(x)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(a).to(torch.device("cuda", 0))


def _prepare(eng, keys):
    """list of 192-byte records -> (device tensor holding the tables, its pointer)"""
    import torch
    n = len(keys)
    d_k = _dev(b"".join(keys))
    tab = torch.empty(n * eng.G2_PREPARED_BYTES, dtype=torch.uint8, device=torch.device("cuda", 0))
    eng.g2_prepare_batch_dev(d_k.data_ptr(), n, tab.data_ptr())
    return tab


def _msgs_dev(msgs):
    buf = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
    off = np.zeros(len(msgs) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(m) for m in msgs])
    return _dev(buf), _dev(off)


def test_tables_are_the_references_g2prepared(eng):
    import torch
    xs = P.XORShift(9001)
    keys = [RC.g2_generator()] + [rand_g2(xs) for _ in range(4)]
    # a record that is no curve point at all: G2AffineToPrepared is formulas only, the table must follow them all the same
    keys.append(RC.g2_mul(RC.g2_generator(), (5).to_bytes(32, "big"))[:96] + bytes(95) + b"\x07")
    # ... and a coordinate encoded as a value >= q (FQReprToFQ turns it into 0 upstream, fq.go:49-56): same lines as the oracle's
    keys.append((P.Q + 5).to_bytes(48, "big") + keys[1][48:])
    want = np.stack([RC.g2_prepare(k) for k in keys])
    got = eng.g2_prepare_batch(b"".join(keys), len(keys))
    assert got.shape == want.shape and np.array_equal(got, want)
    tab = _prepare(eng, keys)
    out = torch.empty(len(keys) * 68 * 3 * 12, dtype=torch.int64, device=torch.device("cuda", 0))
    eng.g2_prepared_export_dev(tab.data_ptr(), len(keys), out.data_ptr())
    assert np.array_equal(out.cpu().numpy().view(np.uint64).reshape(want.shape), want)
    # the start-up table of the generator is the same object
    assert np.array_equal(eng.debug_g2_prepare(mode=2).reshape(68, 3, 12), want[0])


@pytest.mark.parametrize("n", [37, 9000])
def test_pairing_with_prepared_keys_is_the_pairing(eng, n):
    import torch
    xs = P.XORShift(9100 + n)
    nk = 5
    keys = [rand_g2(xs) for _ in range(nk)]
    base = [rand_g1(xs) for _ in range(16)]
    g1 = [base[(7 * i + i // 16) % 16] for i in range(n)]
    idx = np.array([(3 * i + i // 7) % nk for i in range(n)], dtype=np.uint32)
    tab = _prepare(eng, keys)
    d_g1, d_idx = _dev(b"".join(g1)), _dev(idx)
    out = torch.empty(n * 72, dtype=torch.int64, device=torch.device("cuda", 0))
    eng.pairing_batch_prepared_dev(d_g1.data_ptr(), tab.data_ptr(), d_idx.data_ptr(), out.data_ptr(), n)
    got = out.cpu().numpy().view(np.uint64).reshape(n, 72)
    ref = eng.pairing_batch(b"".join(g1), b"".join(keys[j] for j in idx), n)
    assert np.array_equal(got, ref)
    for i in (0, 1, n // 2, n - 1):
        assert np.array_equal(got[i], RC.pairing_batch(g1[i], keys[idx[i]], 1)[0])
    # no index array: tuple t uses table t
    m = min(n, nk)
    out2 = torch.empty(m * 72, dtype=torch.int64, device=torch.device("cuda", 0))
    eng.pairing_batch_prepared_dev(d_g1.data_ptr(), tab.data_ptr(), 0, out2.data_ptr(), m)
    assert np.array_equal(out2.cpu().numpy().view(np.uint64).reshape(m, 72), eng.pairing_batch(b"".join(g1[:m]), b"".join(keys[:m]), m))


def _verify_case(eng, n, seed, nk=6):
    xs = P.XORShift(seed)
    sks = [sk_bytes(xs) for _ in range(nk)]
    keys = [RC.g2pubs.priv_to_pub(s) for s in sks]
    msgs = [b"prepared %d/%d" % (i, seed) + bytes(i % 4) for i in range(n)]
    idx = np.array([(5 * i + i // 3) % nk for i in range(n)], dtype=np.uint32)
    h = eng.hash_g1_batch(msgs)
    sigs, _ = eng.g1_mul_batch(h.reshape(-1), b"".join(sks[j] for j in idx), n)
    return sks, keys, msgs, idx, sigs


@pytest.mark.parametrize("n", [45, 9100])
def test_verify_with_prepared_keys_matches_the_unprepared_verdicts(eng, n):
    import torch
    sks, keys, msgs, idx, sigs = _verify_case(eng, n, 9200 + n)
    keys = keys + [bytes(192)]                                             # table 6: the point at infinity (the reference panics; verdict 0)
    idx = idx.copy(); sigs = sigs.copy()
    expect = np.ones(n, dtype=bool)
    for i in range(3, n, 11):                                              # wrong key
        idx[i] = (idx[i] + 1) % 6; expect[i] = False
    for i in range(5, n, 13):                                              # signature of another message
        sigs[i] = sigs[(i + 1) % n]; expect[i] = False
    idx[7] = 6; expect[7] = False                                          # key at infinity
    sigs[9] = 0; expect[9] = False                                         # signature at infinity (all-zero record)
    inf = np.zeros(n, dtype=np.uint8); inf[12] = 1; expect[12] = False     # flagged by the caller
    tab = _prepare(eng, keys)
    d_m, d_o = _msgs_dev(msgs)
    d_s, d_i, d_f = _dev(sigs.reshape(-1)), _dev(idx), _dev(inf)
    ok = torch.zeros(n, dtype=torch.uint8, device=torch.device("cuda", 0))
    eng.g2pubs_verify_batch_prepared_dev(d_m.data_ptr(), d_o.data_ptr(), tab.data_ptr(), d_i.data_ptr(), d_s.data_ptr(), d_f.data_ptr(), ok.data_ptr(), n)
    got = ok.cpu().numpy().astype(bool)
    allk = b"".join(keys[j] for j in idx)
    ref, _ = eng.g2pubs_verify_batch(msgs, allk, sigs.reshape(-1), inf_flags=inf)
    assert np.array_equal(got, ref.astype(bool))
    assert np.array_equal(got, expect)
    for i in (0, 3, 5, 14, n - 1):                                         # ... and the oracle on a sample (finite inputs only)
        assert RC.g2pubs.verify(msgs[i], keys[idx[i]], sigs[i].tobytes()) == bool(expect[i])


@pytest.mark.parametrize("n", [9, 9001])
def test_verify_aggregate_with_prepared_keys(eng, n):
    sks, keys, msgs, idx, sigs = _verify_case(eng, n, 9300 + n)
    agg = eng.g1_sum(sigs.reshape(-1), n)
    tab = _prepare(eng, keys + [bytes(192)])

    def prepared(ms, ix, sig):
        d_m, d_o = _msgs_dev(ms)
        d_i = _dev(np.asarray(ix, dtype=np.uint32))
        return eng.g2pubs_verify_aggregate_prepared_dev(d_m.data_ptr(), d_o.data_ptr(), tab.data_ptr(), d_i.data_ptr(), sig, len(ms))

    def plain(ms, ix, sig):
        return eng.g2pubs_verify_aggregate(ms, b"".join((keys + [bytes(192)])[j] for j in ix), sig)

    assert prepared(msgs, idx, agg) is True and plain(msgs, idx, agg) is True
    if n < 50:
        assert RC.g2pubs.verify_aggregate(agg, [keys[j] for j in idx], msgs) is True
    bad = idx.copy(); bad[n // 2] = (bad[n // 2] + 1) % 6
    assert prepared(msgs, bad, agg) is False and plain(msgs, bad, agg) is False
    dup = list(msgs); dup[n - 1] = dup[1]
    assert prepared(dup, idx, agg) is False                                # duplicate message (g2pubs/bls.go:245-261)
    infk = idx.copy(); infk[n - 2] = 6
    assert prepared(msgs, infk, agg) is False and plain(msgs, infk, agg) is False
    assert prepared(msgs, idx, bytes(96)) is False
    # odd / even counts around the two-tuples-per-lane-pair split
    for m in (n - 1, n - 2):
        sub = eng.g1_sum(sigs[:m].reshape(-1), m)
        assert prepared(msgs[:m], idx[:m], sub) is True
        assert prepared(msgs[:m], idx[:m], agg) is False


def test_prepared_full_size_batch_against_the_unprepared_kernels():
    """65 536 tuples over 1 024 prepared keys on the throughput kernels: verdict bytes equal those of the unprepared path, with
    every 97th tuple corrupted."""
    import torch
    from bls_amd import engine as eng
    eng.init(0)
    n, nk = 65536, 1024
    sk = b"".join(hashlib.sha256(b"prep-full-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
    pks, _ = eng.g2_mul_generator_batch(sk, nk)
    msgs = [hashlib.sha256(b"m%d" % i).digest() for i in range(n)]
    idx = (np.arange(n, dtype=np.uint64) * 2654435761 % nk).astype(np.uint32)
    skb = np.frombuffer(sk, dtype=np.uint8).reshape(nk, 32)
    h = eng.hash_g1_batch(eng.PackedMsgs(msgs))
    sigs, _ = eng.g1_mul_batch(h.reshape(-1), skb[idx].reshape(-1), n)
    expect = np.ones(n, dtype=bool)
    bad = idx.copy()
    for i in range(11, n, 97):
        bad[i] = (bad[i] + 1) % nk; expect[i] = False
    dev = torch.device("cuda", 0)
    tab = torch.empty(nk * eng.G2_PREPARED_BYTES, dtype=torch.uint8, device=dev)
    d_k = _dev(pks.reshape(-1))
    eng.g2_prepare_batch_dev(d_k.data_ptr(), nk, tab.data_ptr())
    d_m, d_o = _msgs_dev(msgs)
    d_s, d_i = _dev(sigs.reshape(-1)), _dev(bad)
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.g2pubs_verify_batch_prepared_dev(d_m.data_ptr(), d_o.data_ptr(), tab.data_ptr(), d_i.data_ptr(), d_s.data_ptr(), 0, ok.data_ptr(), n)
    got = ok.cpu().numpy().astype(bool)
    assert np.array_equal(got, expect)
    ref, _ = eng.g2pubs_verify_batch(eng.PackedMsgs(msgs), pks[bad].reshape(-1), sigs.reshape(-1))
    assert np.array_equal(ref.astype(bool), got)


def test_library_owned_tables_and_the_host_form(eng):
    """blsmi_g2_prepared_create / _destroy and the host-buffer verify (what the cgo shim calls): same verdicts as the unprepared
    host form, with and without an index array."""
    n = 300
    sks, keys, msgs, idx, sigs = _verify_case(eng, n, 9400)
    pk = eng.PreparedKeys(b"".join(keys), len(keys))
    bad = idx.copy(); bad[17] = (bad[17] + 2) % 6
    ok, bm = eng.g2pubs_verify_batch_prepared(msgs, pk, bad, sigs.reshape(-1))
    ref, rbm = eng.g2pubs_verify_batch(msgs, b"".join(keys[j] for j in bad), sigs.reshape(-1))
    assert np.array_equal(ok, ref) and np.array_equal(bm, rbm) and not ok[17] and ok.sum() == n - 1
    # tuple t uses table t
    ok2, _ = eng.g2pubs_verify_batch_prepared(msgs[:6], pk, None, sigs[:6].reshape(-1))
    ref2, _ = eng.g2pubs_verify_batch(msgs[:6], b"".join(keys), sigs[:6].reshape(-1))
    assert np.array_equal(ok2, ref2)
    # VerifyAggregate over the same tables, host buffers
    agg = eng.g1_sum(sigs.reshape(-1), n)
    assert eng.g2pubs_verify_aggregate_prepared(msgs, pk, idx, agg) is True
    assert eng.g2pubs_verify_aggregate_prepared(msgs, pk, bad, agg) is False
    dup = list(msgs); dup[5] = dup[6]
    assert eng.g2pubs_verify_aggregate_prepared(dup, pk, idx, agg) is False
    assert eng.g2pubs_verify_aggregate_prepared(msgs[:6], pk, None, eng.g1_sum(sigs[:6].reshape(-1), 6)) is (list(idx[:6]) == list(range(6)))
    pk.close()
    pk.close()                                                             # idempotent


def test_host_mirror_prepare_keys(eng):
    """bls_amd.g2pubs.PrepareKeys / VerifyBatchPrepared against VerifyBatch and the oracle, a key at infinity among them"""
    from bls_amd import g2pubs as G
    xs = P.XORShift(9500)
    sks = [sk_bytes(xs) for _ in range(4)]
    pubs = [G.PrivToPub(s) for s in sks] + [G.NewAggregatePubkey()]       # the last one: the zero point
    msgs = [b"mirror %d" % i for i in range(10)]
    idx = [i % 4 for i in range(10)]
    sigs = [G.Sign(m, sks[j]) for m, j in zip(msgs, idx)]
    keys = G.PrepareKeys(pubs)
    assert G.VerifyBatchPrepared(msgs, keys, idx, sigs) == [True] * 10 == G.VerifyBatch(msgs, [pubs[j] for j in idx], sigs)
    idx2 = list(idx); idx2[3] = (idx2[3] + 1) % 4; idx2[6] = 4
    got = G.VerifyBatchPrepared(msgs, keys, idx2, sigs)
    assert got == [i not in (3, 6) for i in range(10)] == G.VerifyBatch(msgs, [pubs[j] for j in idx2], sigs)
    assert RC.g2pubs.verify(msgs[3], pubs[idx2[3]].p.bytes_or_zero(), sigs[3].s.bytes_or_zero()) is False
    with pytest.raises(IndexError):
        G.VerifyBatchPrepared(msgs, keys, [5] * 10, sigs)
    keys.Close()



def test_concurrent_callers_share_one_set_of_prepared_keys(eng):
    """Six threads verify against the same library-owned tables at once (each call leases its own stream context; the tables are only
    read), two more run unprepared batches beside them; every caller gets the verdicts of its own tuples."""
    import threading
    sks, keys, msgs, idx, sigs = _verify_case(eng, 96, 9600)
    pk = eng.PreparedKeys(b"".join(keys), len(keys))
    expect = np.ones(96, dtype=bool)
    bad = idx.copy()
    for i in range(2, 96, 9):
        bad[i] = (bad[i] + 1) % 6; expect[i] = False
    plain_keys = b"".join(keys[j] for j in bad)
    results = {}

    def run(t):
        try:
            for _ in range(4):
                if t < 6:
                    lo = 16 * t
                    ok, _ = eng.g2pubs_verify_batch_prepared(msgs[lo:lo + 16], pk, bad[lo:lo + 16], sigs[lo:lo + 16].reshape(-1))
                    assert np.array_equal(ok, expect[lo:lo + 16])
                else:
                    ok, _ = eng.g2pubs_verify_batch(msgs, plain_keys, sigs.reshape(-1))
                    assert np.array_equal(ok, expect)
            results[t] = True
        except Exception as e:  # noqa: BLE001
            results[t] = e

    th = [threading.Thread(target=run, args=(t,)) for t in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    pk.close()
    assert all(results.get(t) is True for t in range(8)), results
