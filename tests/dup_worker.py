"""Worker of tests/test_gpu_round3.py::test_device_duplicate_table_and_its_fallback: own process so that BLSMI_DUP_FORCE_SORT (read by
the library) can differ per run.  VerifyAggregate over n > 4096 messages -- the duplicate rejection then runs on the device (k_util.hip),
or, with the environment hook, through its host fallback -- for resident and host buffers.  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from bls_amd import engine as eng
    from test_gpu_round3 import _aggregate_case, _dev
    eng.init(0)
    ck = {}
    for group in ("g2pubs", "g1pubs"):
        n = 5000
        msgs, pks, agg, _ = _aggregate_case(group, n, 4242)
        host = eng.g2pubs_verify_aggregate if group == "g2pubs" else eng.g1pubs_verify_aggregate

        def dev_call(ms):
            buf = np.frombuffer(b"".join(ms) or b"\0", dtype=np.uint8)
            off = np.zeros(len(ms) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(m) for m in ms])
            d_m, d_o, d_k = _dev(buf), _dev(off), _dev(b"".join(pks))
            return eng.verify_aggregate_dev(group, d_m.data_ptr(), d_o.data_ptr(), d_k.data_ptr(), agg, len(ms))
        ck[group + "_true"] = dev_call(msgs) is True and host(msgs, b"".join(pks), agg) is True
        dup = list(msgs); dup[n - 1] = dup[7]
        ck[group + "_dup"] = dev_call(dup) is False and host(dup, b"".join(pks), agg) is False
        emp = list(msgs); emp[1234] = b""
        ck[group + "_empty"] = dev_call(emp) is False and host(emp, b"".join(pks), agg) is False
        many = [msgs[0]] * n                                                 # every message equal: one long run of collisions, first comparison hits
        ck[group + "_all_equal"] = dev_call(many) is False and host(many, b"".join(pks), agg) is False
    print("DUP_RESULT " + json.dumps({"forced": bool(os.environ.get("BLSMI_DUP_FORCE_SORT")), "checks": ck, "ok": all(ck.values())}), flush=True)


if __name__ == "__main__":
    main()
