"""Worker of tests/test_gpu_round2.py::test_inlibrary_split_*: runs in its own process because the split of the library
(devices, logical shards, RCCL) is fixed when the library initialises.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from bls_amd import engine as eng
    from gpu_common import P, RC, rand_g1, rand_g2, sk_bytes
    eng.init_devices(1)
    out = {"devices": eng.device_count(), "shards": eng.shard_count(), "version": eng.version(), "checks": {}}
    ck = out["checks"]
    xs = P.XORShift(123)
    # --- pairing batch, split: bit-exact vs the oracle
    n = 200
    g1 = b"".join(rand_g1(xs) for _ in range(n)); g2 = b"".join(rand_g2(xs) for _ in range(n))
    ck["pairing"] = bool(np.array_equal(eng.pairing_batch(g1, g2, n), RC.pairing_batch(g1, g2, n)))
    # --- verify batch, split: verdict bytes and the bitmap (all-reduce path)
    for grp, o in (("g2pubs", RC.g2pubs), ("g1pubs", RC.g1pubs)):
        n = 330                                              # not a multiple of 64: ragged last shard
        msgs, pks, sigs, expect = [], [], [], []
        for i in range(n):
            sk = sk_bytes(xs)
            m = b"Hello world! 16 characters %d" % i
            pk, sig = o.priv_to_pub(sk), o.sign(m, sk)
            good = i % 16 != 15
            if not good:
                if (i // 16) % 2:
                    m += b"!"
                else:
                    pk = o.priv_to_pub(sk_bytes(xs))
            msgs.append(m); pks.append(pk); sigs.append(sig); expect.append(good)
        fn = eng.g2pubs_verify_batch if grp == "g2pubs" else eng.g1pubs_verify_batch
        ok, bitmap = fn(msgs, b"".join(pks), b"".join(sigs))
        ck[grp + "_verdicts"] = list(ok) == expect
        ck[grp + "_bitmap"] = [bool(bitmap[i >> 3] >> (i & 7) & 1) for i in range(n)] == expect and len(bitmap) == (n + 7) // 8
        ck[grp + "_oracle_sample"] = all(o.verify(msgs[i], pks[i], sigs[i]) == expect[i] for i in (0, 15, 31, 64, 65, 200, n - 1))
        flags = np.zeros(n, dtype=np.uint8); flags[70] = 1; flags[300] = 2
        ok, _ = fn(msgs, b"".join(pks), b"".join(sigs), flags)
        exp2 = list(expect); exp2[70] = False; exp2[300] = False
        ck[grp + "_flags"] = list(ok) == exp2
        # --- one n-way VerifyAggregate, split: partial products gathered, single final exponentiation
        n2 = 200
        sks = [sk_bytes(xs) for _ in range(n2)]
        ms = [b"distinct message %d" % i for i in range(n2)]
        pk2 = [o.priv_to_pub(sk) for sk in sks]
        agg = (RC.g1_sum if grp == "g2pubs" else RC.g2_sum)(b"".join(o.sign(m, sk) for m, sk in zip(ms, sks)), n2)
        va = eng.g2pubs_verify_aggregate if grp == "g2pubs" else eng.g1pubs_verify_aggregate
        ck[grp + "_aggregate_true"] = va(ms, b"".join(pk2), agg) is True
        sw = list(pk2); sw[150], sw[151] = sw[151], sw[150]
        ck[grp + "_aggregate_swapped"] = va(ms, b"".join(sw), agg) is False
        dup = list(ms); dup[199] = dup[3]
        ck[grp + "_aggregate_duplicate"] = va(dup, b"".join(pk2), agg) is False
        infk = list(pk2); infk[180] = bytes(len(pk2[0]))
        ck[grp + "_aggregate_inf_key"] = va(ms, b"".join(infk), agg) is False
        ck[grp + "_aggregate_small_oracle"] = va(ms[:5], b"".join(pk2[:5]), agg) == o.verify_aggregate(agg, pk2[:5], ms[:5])
        small = (RC.g1_sum if grp == "g2pubs" else RC.g2_sum)(b"".join(o.sign(m, sk) for m, sk in zip(ms[:5], sks[:5])), 5)
        ck[grp + "_aggregate_small_true"] = va(ms[:5], b"".join(pk2[:5]), small) is True and o.verify_aggregate(small, pk2[:5], ms[:5]) is True
    # --- scalar multiplication batches are split the same way
    n3 = 300
    pts = b"".join(rand_g1(xs) for _ in range(8)) * 38
    ks = [sk_bytes(xs) for _ in range(n3)]
    o3, i3 = eng.g1_mul_batch(pts[:96 * n3], b"".join(ks), n3)
    ck["g1_mul_split"] = (not i3.any()) and all(o3[i].tobytes() == RC.g1_mul(pts[96 * i:96 * i + 96], ks[i]) for i in (0, 1, 99, 100, 199, 200, 299))
    out["dup_screen"] = "sort" if os.environ.get("BLSMI_DUP_FORCE_SORT") else "hash table"
    out["ok"] = all(ck.values())
    eng.shutdown()
    print("SHARD_WORKER_RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
