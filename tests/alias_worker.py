"""Worker of tests/test_gpu_alias.py: own process because the device list is fixed when the library initialises.
BLSMI_DEVICE_ALIAS=0,0[,0,0] gives N logical devices on this one GPU (include/blsmi.h): g_dev[d] for d > 0, the shard -> device
routing, per-device context pools / generator tables / exchange buffers, the bitmap all-reduce and the partial-product all-gather
(host-staged stand-ins under the hook) and the owner routing of the *_dev entry points all execute.  Prints one JSON line."""
import hashlib
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ndev = int(sys.argv[1])
    n_agg = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    import torch
    from bls_amd import engine as eng
    from gpu_common import RC
    eng.init_devices(0)
    out = {"devices": eng.device_count(), "shards": eng.shard_count(), "version": eng.version(), "checks": {}}
    ck = out["checks"]
    ck["device_count"] = eng.device_count() == ndev and eng.shard_count() == ndev and "ALIASED-DEVICES" in eng.version()

    def leases():
        return [eng.device_leases(d) for d in range(ndev)]
    nk = 256
    sk = b"".join(hashlib.sha256(b"alias-sk-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
    pks, _ = eng.g2_mul_generator_batch(sk, nk)
    # ---- one n-signature g2pubs VerifyAggregate, ndev shards (one per logical device), partial products gathered
    n = n_agg
    msgs = [hashlib.sha256(int(i).to_bytes(8, "little")).digest() for i in range(n)]
    packed = eng.PackedMsgs(msgs)
    h = eng.hash_g1_batch(packed)
    sigs, _ = eng.g1_mul_batch(h.reshape(-1), sk * (n // nk), n)
    agg = eng.g1_sum(sigs.reshape(-1), n)
    # the fused Sign entry point over the same split batch (later shards rebase their message offsets): the two-step signatures, byte for byte;
    # and with ragged messages against the oracle
    fs, _ = eng.g2pubs_sign_batch(packed, sk * (n // nk))
    ck["fused_sign_split_equals_hash_then_multiply"] = bool(np.array_equal(fs, sigs))
    nr = 3 * 8192 + 77
    ragged = [(b"r%d" % i) * (1 + i % 5) for i in range(nr)]
    fr, _ = eng.g1pubs_sign_batch(ragged, (sk * (nr // nk + 1))[:32 * nr])
    ck["fused_sign_ragged_split_oracle_rows"] = all(fr[i].tobytes() == RC.g1pubs.sign(ragged[i], sk[32 * (i % nk):32 * (i % nk) + 32]) for i in (0, 8191, 8192, 16500, nr - 1))
    allpk = np.ascontiguousarray(np.tile(pks, (n // nk, 1))).reshape(-1)
    l0 = leases()
    ck["aggregate_true"] = eng.g2pubs_verify_aggregate(packed, allpk, agg) is True
    l1 = leases()
    ck["aggregate_every_device_served_a_shard"] = all(b > a for a, b in zip(l0, l1))
    per = n // ndev
    i_bad = per + per // 3                                                           # a row of device 1's shard
    bad = allpk.copy(); bad[192 * i_bad:192 * (i_bad + 1)] = pks[(i_bad + 1) % nk]
    ck["aggregate_wrong_key_in_device_1_shard"] = eng.g2pubs_verify_aggregate(packed, bad, agg) is False
    dup = list(msgs); dup[n - 1] = dup[123]                                          # duplicate across the first and the last device
    ck["aggregate_duplicate_across_first_and_last_device"] = eng.g2pubs_verify_aggregate(dup, allpk, agg) is False
    small = eng.g1_sum(sigs[:3].reshape(-1), 3)
    ck["aggregate_small_oracle"] = eng.g2pubs_verify_aggregate(msgs[:3], allpk[:192 * 3], small) is True and RC.g2pubs.verify_aggregate(small, [pks[i].tobytes() for i in range(3)], msgs[:3]) is True
    # the g1pubs mirror at a quarter of the size: swapped-group kernels on every logical device
    n1 = max(ndev * 8192, n // 4)
    pk1, _ = eng.g1_mul_generator_batch(sk, nk)
    h2 = eng.hash_g2_batch(eng.PackedMsgs(msgs[:n1]))
    s2, _ = eng.g2_mul_batch(h2.reshape(-1), sk * (n1 // nk), n1)
    agg2 = eng.g2_sum(s2.reshape(-1), n1)
    allpk1 = np.ascontiguousarray(np.tile(pk1, (n1 // nk, 1))).reshape(-1)
    ck["g1pubs_aggregate_true"] = eng.g1pubs_verify_aggregate(msgs[:n1], allpk1, agg2) is True
    bad1 = allpk1.copy(); j = n1 - 5; bad1[96 * j:96 * (j + 1)] = pk1[(j + 1) % nk]
    ck["g1pubs_aggregate_wrong_key_in_last_shard"] = eng.g1pubs_verify_aggregate(msgs[:n1], bad1, agg2) is False
    # ---- ndev x 32 768 verifies, verdict bytes + bitmap (all-reduce of the packed bits), every 4099th tuple corrupted
    nv = ndev * 32768
    vm = msgs[:nv]; vs = sigs[:nv].copy(); vpk = allpk[:192 * nv].reshape(nv, 192).copy()
    expect = np.ones(nv, dtype=bool)
    for i in range(17, nv, 4099):
        vpk[i] = pks[(i + 3) % nk]; expect[i] = False
    l0 = leases()
    ok, bitmap = eng.g2pubs_verify_batch(eng.PackedMsgs(vm), vpk.reshape(-1), vs.reshape(-1))
    l1 = leases()
    ck["verify_split_every_device"] = all(b > a for a, b in zip(l0, l1))
    ck["verify_split_verdicts"] = bool(np.array_equal(ok, expect))
    bits = np.unpackbits(bitmap, bitorder="little")[:nv].astype(bool)
    ck["verify_split_bitmap"] = bool(np.array_equal(bits, expect)) and len(bitmap) == nv // 8
    ck["verify_oracle_sample"] = all(RC.g2pubs.verify(vm[i], vpk[i].tobytes(), vs[i].tobytes()) == bool(expect[i]) for i in (0, 17, 32767, 32768, 17 + 4099, nv - 1))
    # ---- the same split calls over the reference's in-memory points (blsmi 0.6, *_jac): Jacobian records with z != 1 are cut at the
    # 288 / 144-byte record boundaries, every shard runs ToAffine on its own device; verdicts and bitmap identical to the affine call
    from gpu_common import P, jac1, jac2
    xs = P.XORShift(9090)
    pkj256 = [jac2(xs, pks[i].tobytes()) for i in range(nk)]
    sgj_rows = 2048
    sgj = [jac1(xs, vs[i].tobytes()) for i in range(sgj_rows)]                       # (2 048 distinct signatures; the rest of the batch repeats tuples)
    idx = np.arange(nv) % sgj_rows
    jm = eng.PackedMsgs([vm[i] for i in idx])
    jpk = np.frombuffer(b"".join(pkj256), dtype=np.uint8).reshape(nk, 288)[idx % nk]
    jex = np.ones(nv, dtype=bool)
    jpk = np.ascontiguousarray(jpk)
    for i in range(29, nv, 5003):
        jpk[i] = np.frombuffer(pkj256[(idx[i] + 7) % nk], dtype=np.uint8); jex[i] = False
    jsg = np.frombuffer(b"".join(sgj), dtype=np.uint8).reshape(sgj_rows, 144)[idx]
    l0 = leases()
    okj, bmj = eng.g2pubs_verify_batch_jac(jm, jpk.reshape(-1), np.ascontiguousarray(jsg).reshape(-1))
    l1 = leases()
    ck["jac_verify_split_every_device"] = all(b > a for a, b in zip(l0, l1))
    ck["jac_verify_split_verdicts_and_bitmap"] = bool(np.array_equal(okj, jex)) and bool(np.array_equal(np.unpackbits(bmj, bitorder="little")[:nv].astype(bool), jex))
    na = ndev * 16384
    aggj = jac1(xs, eng.g1_sum(sigs[:na].reshape(-1), na))
    apkj = np.ascontiguousarray(np.frombuffer(b"".join(pkj256), dtype=np.uint8).reshape(nk, 288)[np.arange(na) % nk])
    ck["jac_aggregate_split_true"] = eng.g2pubs_verify_aggregate_jac(msgs[:na], apkj.reshape(-1), aggj) is True
    apkj[na - 3] = np.frombuffer(pkj256[5], dtype=np.uint8)
    ck["jac_aggregate_split_wrong_key_in_last_shard"] = eng.g2pubs_verify_aggregate_jac(msgs[:na], apkj.reshape(-1), aggj) is False
    # blsmi_trim from another thread WHILE split calls run: the exchange buffers belong to the split call; verdicts unchanged
    stop = threading.Event()

    def trimmer():
        while not stop.is_set():
            eng.trim(0)
    tt = threading.Thread(target=trimmer); tt.start()
    try:
        good = True
        for _ in range(3):
            okt, bmt = eng.g2pubs_verify_batch(eng.PackedMsgs(vm), vpk.reshape(-1), vs.reshape(-1))
            good = good and bool(np.array_equal(okt, expect)) and bool(np.array_equal(np.unpackbits(bmt, bitorder="little")[:nv].astype(bool), expect))
            good = good and eng.g2pubs_verify_aggregate(msgs[:na], allpk[:192 * na], eng.g1_sum(sigs[:na].reshape(-1), na)) is True
    finally:
        stop.set(); tt.join()
    ck["trim_during_split_calls"] = good
    # ---- pairings split ndev ways: every shard's first, last and a middle row against the oracle
    npair = ndev * 8192
    g1 = sigs[:npair]; g2 = allpk[:192 * npair].reshape(npair, 192)
    po = eng.pairing_batch(g1.reshape(-1), g2.reshape(-1), npair)
    rows = sorted({d * 8192 + o for d in range(ndev) for o in (0, 4097, 8191)})
    ck["pairing_split_oracle_rows"] = all(np.array_equal(po[i], RC.pairing_batch(g1[i].tobytes(), g2[i].tobytes(), 1)[0]) for i in rows)
    # ---- *_dev entry points run on the logical device that OWNS the caller's buffers
    dev = torch.device("cuda", 0)
    m = 8192 + 64
    want = po[:m]
    routed = []
    for d in range(ndev):
        a = torch.from_numpy(np.ascontiguousarray(g1[:m])).to(dev); b = torch.from_numpy(np.ascontiguousarray(g2[:m])).to(dev)
        o = torch.zeros((m, 72), dtype=torch.int64, device=dev)
        for t in (a, b, o):
            eng.debug_alias_own(t.data_ptr(), t.numel() * t.element_size(), d)
        l0 = leases()
        eng.pairing_batch_dev(a.data_ptr(), b.data_ptr(), o.data_ptr(), m)
        l1 = leases()
        routed.append([y - x for x, y in zip(l0, l1)] == [1 if e == d else 0 for e in range(ndev)] and bool(np.array_equal(o.cpu().numpy().view(np.uint64), want)))
        # resident verify on the same owner
        dm = torch.from_numpy(packed.buf[:32 * m].copy()).to(dev); do = torch.from_numpy(packed.off[:m + 1].view(np.int64).copy()).to(dev)
        dk = torch.from_numpy(np.ascontiguousarray(vpk[:m])).to(dev); dsg = torch.from_numpy(np.ascontiguousarray(vs[:m])).to(dev); dok = torch.zeros(m, dtype=torch.uint8, device=dev)
        for t in (dm, do, dk, dsg, dok):
            eng.debug_alias_own(t.data_ptr(), t.numel() * t.element_size(), d)
        l0 = leases()
        eng.verify_batch_dev("g2pubs", dm.data_ptr(), do.data_ptr(), dk.data_ptr(), dsg.data_ptr(), 0, dok.data_ptr(), m)
        l1 = leases()
        routed.append([y - x for x, y in zip(l0, l1)] == [1 if e == d else 0 for e in range(ndev)] and bool(np.array_equal(dok.cpu().numpy().astype(bool), expect[:m])))
        for t in (a, b, o, dm, do, dk, dsg, dok):
            eng.debug_alias_own(t.data_ptr(), 0, d)
    ck["dev_forms_routed_by_buffer_owner"] = all(routed)
    # ---- concurrent callers with small (unsplit) calls spread over the logical devices
    res = [None] * 12
    l0 = leases()

    def caller(t):
        lo = 100 * t
        ok_t, _ = eng.g2pubs_verify_batch(eng.PackedMsgs(vm[lo:lo + 100]), vpk[lo:lo + 100].reshape(-1), vs[lo:lo + 100].reshape(-1))
        p_t = eng.pairing_batch(g1[lo:lo + 50].reshape(-1), g2[lo:lo + 50].reshape(-1), 50)
        res[t] = bool(np.array_equal(ok_t, expect[lo:lo + 100])) and bool(np.array_equal(p_t, po[lo:lo + 50]))
    for _ in range(3):
        th = [threading.Thread(target=caller, args=(t,)) for t in range(12)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    l1 = leases()
    ck["concurrent_callers_correct"] = all(res)
    ck["concurrent_callers_reach_every_device"] = all(b > a for a, b in zip(l0, l1))
    out["leases"] = leases()
    out["ok"] = all(ck.values())
    eng.shutdown()
    print("ALIAS_RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
