"""Worker of tests/test_gpu_alias.py::test_one_million_signature_aggregate_over_eight_logical_devices (own process: the device list is fixed when
the library initialises).  BLSMI_DEVICE_ALIAS=0,0,0,0,0,0,0,0: EIGHT logical devices on this one GPU -- the shape of BASELINE configs[3] --
and one 2^20-signature g2pubs VerifyAggregate handed over as the reference's in-memory points (blsmi_g2pubs_verify_aggregate_jac: the entry
the Go shim calls), so that the first real 8-GPU run exercises nothing new but xGMI.  Prints one JSON line."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ndev = int(sys.argv[1]); n = int(sys.argv[2])
    from bls_amd import engine as eng
    from gpu_common import P, RC, jac1, jac2
    eng.init_devices(0)
    out = {"devices": eng.device_count(), "shards": eng.shard_count(), "version": eng.version(), "checks": {}}
    ck = out["checks"]
    ck["device_count"] = eng.device_count() == ndev and eng.shard_count() == ndev and "ALIASED-DEVICES" in eng.version()
    nk = 256
    sk = b"".join(hashlib.sha256(b"alias8-sk-%d" % i).digest()[:31].rjust(32, b"\0") for i in range(nk))
    pks, _ = eng.g2_mul_generator_batch(sk, nk)
    msgs = [hashlib.sha256(b"a8" + int(i).to_bytes(8, "little")).digest() for i in range(n)]
    packed = eng.PackedMsgs(msgs)
    sigs, _ = eng.g2pubs_sign_batch(packed, sk * (n // nk))                 # (itself split ndev ways)
    agg = eng.g1_sum(sigs.reshape(-1), n)
    xs = P.XORShift(8080)
    pkj = np.frombuffer(b"".join(jac2(xs, pks[i].tobytes()) for i in range(nk)), dtype=np.uint8).reshape(nk, 288)     # random z per key
    allpk = np.ascontiguousarray(pkj[np.arange(n) % nk])
    aggj = jac1(xs, agg)

    def leases():
        return [eng.device_leases(d) for d in range(ndev)]
    l0 = leases()
    ck["jac_aggregate_true"] = eng.g2pubs_verify_aggregate_jac(packed, allpk.reshape(-1), aggj) is True
    l1 = leases()
    ck["every_device_served_a_shard"] = all(b > a for a, b in zip(l0, l1))
    per = n // ndev
    for d in (0, ndev // 2, ndev - 1):                                      # a wrong key in the first, a middle and the last device's shard
        bad = allpk.copy(); i = d * per + per // 3; bad[i] = pkj[(i + 1) % nk]
        ck["wrong_key_in_shard_%d" % d] = eng.g2pubs_verify_aggregate_jac(packed, bad.reshape(-1), aggj) is False
    dup = list(msgs); dup[n - 1] = dup[7]                                   # the same message on the first and the last device
    ck["duplicate_across_first_and_last_device"] = eng.g2pubs_verify_aggregate_jac(dup, allpk.reshape(-1), aggj) is False
    # a prefix small enough for the oracle, same keys / signatures: the verdict is the reference's
    m = 5
    small = eng.g1_sum(sigs[:m].reshape(-1), m)
    ck["oracle_prefix"] = RC.g2pubs.verify_aggregate(small, [pks[i].tobytes() for i in range(m)], msgs[:m]) is True and \
        eng.g2pubs_verify_aggregate_jac(msgs[:m], allpk[:m].reshape(-1), jac1(xs, small)) is True
    out["ok"] = all(ck.values())
    print("ALIAS_RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
