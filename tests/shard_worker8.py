"""Worker of tests/test_gpu_round3.py::test_eight_logical_shards_full_size_on_one_gpu: own process because the split of the
library (devices, logical shards, RCCL) is fixed when the library initialises.  BLSMI_SHARDS=8 on one GPU walks every
shard size, slot index and merge the 8-GPU run of BASELINE configs[3] will see.  Prints one JSON line."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from bls_amd import engine as eng
    from gpu_common import RC
    eng.init_devices(1)
    out = {"devices": eng.device_count(), "shards": eng.shard_count(), "version": eng.version(), "checks": {}}
    ck = out["checks"]
    nk = 256
    sk = b"".join(hashlib.sha256(b"shard8-sk-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
    pks, _ = eng.g2_mul_generator_batch(sk, nk)
    # ---- configs[3]: one 2^20-signature g2pubs VerifyAggregate, 8 shards of 131 072
    n = 1 << 20
    msgs = [hashlib.sha256(int(i).to_bytes(8, "little")).digest() for i in range(n)]
    packed = eng.PackedMsgs(msgs)
    h = eng.hash_g1_batch(packed_list(msgs))
    sigs, _ = eng.g1_mul_batch(h.reshape(-1), sk * (n // nk), n)
    agg = eng.g1_sum(sigs.reshape(-1), n)
    allpk = np.ascontiguousarray(np.tile(pks, (n // nk, 1))).reshape(-1)
    ck["aggregate_1m_true"] = eng.g2pubs_verify_aggregate(packed, allpk, agg) is True
    bad = allpk.copy(); bad[192 * 777777:192 * 777778] = pks[(777777 + 1) % nk]      # one wrong key in shard 5
    ck["aggregate_1m_one_wrong_key"] = eng.g2pubs_verify_aggregate(packed, bad, agg) is False
    dup = list(msgs); dup[n - 1] = dup[123]                                            # duplicate across shards 0 and 7
    ck["aggregate_1m_duplicate_across_shards"] = eng.g2pubs_verify_aggregate(dup, allpk, agg) is False
    # a prefix whose oracle verdict is computable: 3 tuples (unsplit, same process)
    small = eng.g1_sum(sigs[:3].reshape(-1), 3)
    ck["aggregate_small_oracle"] = eng.g2pubs_verify_aggregate(msgs[:3], allpk[:192 * 3], small) is True and RC.g2pubs.verify_aggregate(small, [pks[i].tobytes() for i in range(3)], msgs[:3]) is True
    # ---- 8 x 65 536 verifies, verdict bytes + bitmap (all-reduce of the packed bits), every 4099th tuple corrupted
    nv = 8 * 65536
    vm = msgs[:nv]
    vs = sigs[:nv].copy()
    vpk = allpk[:192 * nv].reshape(nv, 192).copy()
    expect = np.ones(nv, dtype=bool)
    for i in range(17, nv, 4099):
        vpk[i] = pks[(i + 3) % nk]; expect[i] = False
    ok, bitmap = eng.g2pubs_verify_batch(packed_list(vm), vpk.reshape(-1), vs.reshape(-1))
    ck["verify_8x64k_verdicts"] = bool(np.array_equal(ok, expect))
    bits = np.unpackbits(bitmap, bitorder="little")[:nv].astype(bool)
    ck["verify_8x64k_bitmap"] = bool(np.array_equal(bits, expect)) and len(bitmap) == nv // 8
    ck["verify_oracle_sample"] = all(RC.g2pubs.verify(vm[i], vpk[i].tobytes(), vs[i].tobytes()) == bool(expect[i]) for i in (0, 17, 65535, 65536, 17 + 4099, nv - 1))
    # ---- pairings split 8 ways: a sample against the oracle
    npair = 8 * 8192
    g1 = sigs[:npair]; g2 = allpk[:192 * npair].reshape(npair, 192)
    po = eng.pairing_batch(g1.reshape(-1), g2.reshape(-1), npair)
    ck["pairing_split_oracle_sample"] = all(np.array_equal(po[i], RC.pairing_batch(g1[i].tobytes(), g2[i].tobytes(), 1)[0]) for i in (0, 8191, 8192, 30000, npair - 1))
    out["ok"] = all(ck.values())
    eng.shutdown()
    print("SHARD8_RESULT " + json.dumps(out), flush=True)


def packed_list(msgs):
    from bls_amd import engine as eng
    return eng.PackedMsgs(msgs)


if __name__ == "__main__":
    main()
