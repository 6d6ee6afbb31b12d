"""-m gpu: the multi-device code of the library (g_ndev > 1) on a one-GPU box, through the BLSMI_DEVICE_ALIAS test hook
(include/blsmi.h): N logical devices with their own context pools, streams, generator tables and exchange buffers on physical GPU 0,
the two collectives through a host-staged stand-in (RCCL refuses two ranks per GPU; the product path stays RCCL).  What the reference
offers as one call (g2pubs/bls.go:159, 240) is split over the devices inside the library."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(ndev, n_agg, script="alias_worker.py"):
    env = dict(os.environ)
    env["BLSMI_DEVICE_ALIAS"] = ",".join(["0"] * ndev)
    env.pop("BLSMI_SHARDS", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script), str(ndev), str(n_agg)], env=env, capture_output=True, text=True, timeout=1200)
    line = [l for l in r.stdout.splitlines() if l.startswith("ALIAS_RESULT ")]
    assert r.returncode == 0 and line, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(line[-1][len("ALIAS_RESULT "):])


@pytest.mark.parametrize("ndev,n_agg", [(2, 1 << 18), (4, 1 << 20)])
def test_logical_devices_on_one_gpu(ndev, n_agg):
    """verify_batch + bitmap all-reduce, pairing_batch, the n-signature VerifyAggregate (true / wrong key in device 1's shard /
    duplicate across the first and last device) with its partial-product all-gather, *_dev routing by buffer owner, concurrent
    callers -- at 2 and 4 logical devices; the 4-device case at BASELINE configs[3]'s full 2^20 signatures."""
    res = _worker(ndev, n_agg)
    assert res["devices"] == ndev and res["ok"], res


def test_one_million_signature_aggregate_over_eight_logical_devices():
    """BASELINE configs[3] in its own shape -- 2^20 signatures, EIGHT devices -- through the entry point the Go shim calls
    (blsmi_g2pubs_verify_aggregate_jac, the reference's in-memory points): true; a wrong key in the first, a middle and the last device's shard;
    a duplicate message across the first and the last device; every logical device serves a shard.  What an 8-GPU node adds to this is xGMI."""
    res = _worker(8, 1 << 20, "alias_worker8.py")
    assert res["devices"] == 8 and res["ok"], res


def test_alias_hook_is_off_without_the_variable():
    """blsmi_debug_alias_own is refused outside the hook, and init_devices does not accept more devices than the box has"""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from bls_amd import engine as e, _native\n"
            "import torch\n"
            "lib = _native.load()\n"
            "assert lib.blsmi_init_devices(torch.cuda.device_count() + 1) == -3\n"
            "e.init_devices(1)\n"
            "assert 'ALIASED' not in e.version()\n"
            "assert lib.blsmi_debug_alias_own(1 << 20, 64, 0) == -3\n"
            "print('OK')\n" % ROOT)
    env = dict(os.environ); env.pop("BLSMI_DEVICE_ALIAS", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
