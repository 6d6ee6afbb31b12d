"""Round-3 soak of the prepared-key entry points: random batch sizes on both sides of the latency hand-over, random key populations and
corruptions; verdicts of the prepared forms against the unprepared ones, pairings bit for bit, samples against the oracle.
python tools/soak4.py [seconds]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from bls_amd import engine as eng
from oracle import refcpu as RC
eng.init(0)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "424242")))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
def t(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.uint64: a = a.view(np.int64)
    if a.dtype == np.uint32: a = a.view(np.int32)
    return torch.from_numpy(a).to(dev)
t0 = time.time(); rounds = 0; checked = 0
while time.time() - t0 < budget:
    n = int(rng.choice([rng.integers(1, 300), rng.integers(8100, 8300), rng.integers(9000, 20000)]))
    nk = int(rng.integers(1, 40))
    sk = rng.integers(0, 256, size=(nk, 32), dtype=np.uint8); sk[:, 0] &= 0x3f; sk[:, 31] |= 1
    pks, _ = eng.g2_mul_generator_batch(sk.reshape(-1), nk)
    if nk > 3 and rng.integers(0, 3) == 0: pks[int(rng.integers(0, nk))] = 0                     # a key at infinity
    msgs = [hashlib.sha256(b"s4-%d-%d" % (rounds, i)).digest()[: 1 + i % 32] for i in range(n)]
    idx = rng.integers(0, nk, size=n).astype(np.uint32)
    h = eng.hash_g1_batch(eng.PackedMsgs(msgs))
    sigs, _ = eng.g1_mul_batch(h.reshape(-1), sk[idx].reshape(-1), n)
    bad = idx.copy()
    for j in rng.integers(0, n, size=max(1, n // 50)): bad[j] = (bad[j] + 1) % nk
    for j in rng.integers(0, n, size=max(1, n // 200)): sigs[j] = 0
    tab = torch.empty(nk * eng.G2_PREPARED_BYTES, dtype=torch.uint8, device=dev)
    d_k = t(pks.reshape(-1)); eng.g2_prepare_batch_dev(d_k.data_ptr(), nk, tab.data_ptr())
    buf = np.frombuffer(b"".join(msgs), dtype=np.uint8); off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum([len(m) for m in msgs])
    d_m, d_o, d_i, d_s = t(buf.copy()), t(off), t(bad), t(sigs.reshape(-1))
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.g2pubs_verify_batch_prepared_dev(d_m.data_ptr(), d_o.data_ptr(), tab.data_ptr(), d_i.data_ptr(), d_s.data_ptr(), 0, ok.data_ptr(), n)
    got = ok.cpu().numpy().astype(bool)
    ref, _ = eng.g2pubs_verify_batch(eng.PackedMsgs(msgs), pks[bad].reshape(-1), sigs.reshape(-1))
    assert np.array_equal(got, ref.astype(bool)), ("verify", n, nk)
    for j in rng.integers(0, n, size=2):
        if pks[bad[j]].any() and sigs[j].any():
            assert RC.g2pubs.verify(msgs[j], pks[bad[j]].tobytes(), sigs[j].tobytes()) == bool(got[j]); checked += 1
    # pairings over the same tables (finite keys only)
    fin = np.array([pks[i].any() for i in range(nk)])
    if fin.all():
        m = min(n, 12000)
        o1 = torch.empty(m * 72, dtype=torch.int64, device=dev)
        d_h = t(h[:m].reshape(-1))
        eng.pairing_batch_prepared_dev(d_h.data_ptr(), tab.data_ptr(), d_i.data_ptr(), o1.data_ptr(), m)
        want = eng.pairing_batch(h[:m].reshape(-1), pks[bad[:m]].reshape(-1), m)
        assert np.array_equal(o1.cpu().numpy().view(np.uint64).reshape(m, 72), want), ("pairing", m, nk)
    rounds += 1
print("soak4 ok: %d rounds, %d oracle samples, %.0f s" % (rounds, checked, time.time() - t0))
