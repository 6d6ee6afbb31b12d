"""Prepared public keys against the ordinary path, inputs resident in HBM: 65 536 g2pubs verifies, 65 536 pairings, one 2^20-message
VerifyAggregate, and the cost of preparing.  Prints one JSON line.  python tools/prepared_probe.py [nkeys]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def best(fn, reps=3):
    import torch
    fn(); torch.cuda.synchronize()
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); b = min(b, time.perf_counter() - t0)
    return b


def main():
    import torch
    from bls_amd import engine as eng
    eng.init(0)
    dev = torch.device("cuda", 0)
    nk = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    n = 65536

    def t(a):
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint64: a = a.view(np.int64)
        if a.dtype == np.uint32: a = a.view(np.int32)
        return torch.from_numpy(a).to(dev)

    sk = b"".join(hashlib.sha256(b"probe-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
    skb = np.frombuffer(sk, dtype=np.uint8).reshape(nk, 32)
    pks, _ = eng.g2_mul_generator_batch(sk, nk)
    d_pk = t(pks.reshape(-1))
    tab = torch.empty(nk * eng.G2_PREPARED_BYTES, dtype=torch.uint8, device=dev)
    out = {"keys": nk, "table_MB": round(nk * eng.G2_PREPARED_BYTES / 1e6, 1)}
    tp = best(lambda: eng.g2_prepare_batch_dev(d_pk.data_ptr(), nk, tab.data_ptr()))
    out["prepare_ms"] = round(tp * 1e3, 3); out["prepare_keys_per_s"] = round(nk / tp, 1)

    def case(n):
        msgs = [hashlib.sha256(b"pm%d" % i).digest() for i in range(n)]
        idx = (np.arange(n, dtype=np.uint64) * 2654435761 % nk).astype(np.uint32)
        h = eng.hash_g1_batch(eng.PackedMsgs(msgs))
        sigs, _ = eng.g1_mul_batch(h.reshape(-1), skb[idx].reshape(-1), n)
        buf = np.frombuffer(b"".join(msgs), dtype=np.uint8)
        off = np.arange(n + 1, dtype=np.uint64) * 32
        return msgs, idx, sigs, t(buf), t(off), t(idx), t(sigs.reshape(-1)), t(pks[idx].reshape(-1)), h

    msgs, idx, sigs, d_m, d_o, d_i, d_s, d_allpk, h = case(n)
    ok1 = torch.zeros(n, dtype=torch.uint8, device=dev); ok2 = torch.zeros(n, dtype=torch.uint8, device=dev)
    t_plain = best(lambda: eng.verify_batch_dev("g2pubs", d_m.data_ptr(), d_o.data_ptr(), d_allpk.data_ptr(), d_s.data_ptr(), 0, ok1.data_ptr(), n))
    t_prep = best(lambda: eng.g2pubs_verify_batch_prepared_dev(d_m.data_ptr(), d_o.data_ptr(), tab.data_ptr(), d_i.data_ptr(), d_s.data_ptr(), 0, ok2.data_ptr(), n))
    assert bool(ok1.all()) and bool(ok2.all())
    out["verify_65536"] = {"plain_ms": round(t_plain * 1e3, 3), "prepared_ms": round(t_prep * 1e3, 3),
                           "plain_per_s": round(n / t_plain, 1), "prepared_per_s": round(n / t_prep, 1)}
    # pairings: P = the hashes, Q = the keys
    d_h = t(h.reshape(-1))
    o1 = torch.empty(n * 72, dtype=torch.int64, device=dev); o2 = torch.empty(n * 72, dtype=torch.int64, device=dev)
    p_plain = best(lambda: eng.pairing_batch_dev(d_h.data_ptr(), d_allpk.data_ptr(), o1.data_ptr(), n))
    p_prep = best(lambda: eng.pairing_batch_prepared_dev(d_h.data_ptr(), tab.data_ptr(), d_i.data_ptr(), o2.data_ptr(), n))
    assert torch.equal(o1, o2)
    out["pairing_65536"] = {"plain_ms": round(p_plain * 1e3, 3), "prepared_ms": round(p_prep * 1e3, 3),
                            "plain_per_s": round(n / p_plain, 1), "prepared_per_s": round(n / p_prep, 1)}
    del o1, o2, d_allpk, d_h
    # one 2^20-message aggregate
    N = 1 << 20
    msgs, idx, sigs, d_m, d_o, d_i, d_s, d_allpk, h = case(N)
    agg = eng.g1_sum(sigs.reshape(-1), N)
    res = {}
    a_plain = best(lambda: res.__setitem__("a", eng.verify_aggregate_dev("g2pubs", d_m.data_ptr(), d_o.data_ptr(), d_allpk.data_ptr(), agg, N)), 2)
    a_prep = best(lambda: res.__setitem__("b", eng.g2pubs_verify_aggregate_prepared_dev(d_m.data_ptr(), d_o.data_ptr(), tab.data_ptr(), d_i.data_ptr(), agg, N)), 2)
    assert res["a"] is True and res["b"] is True
    out["aggregate_2^20"] = {"plain_ms": round(a_plain * 1e3, 2), "prepared_ms": round(a_prep * 1e3, 2)}
    print("PREPARED_PROBE " + json.dumps(out))


if __name__ == "__main__":
    main()
