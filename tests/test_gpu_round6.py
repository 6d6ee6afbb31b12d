"""-m gpu, round 6.

* the plain-C client drives the *_jac entry points with C structs laid out like *bls.G1Projective / *bls.G2Projective -- what the Go
  shims hand over since blsmi 0.6 (item 2d): verdicts, sums and signatures against the oracle."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from gpu_common import P, RC, jac1, jac2, sk_bytes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _u64(b):
    return np.frombuffer(b, dtype=np.uint64)


def test_c_abi_jac_leg_from_a_plain_c_client(tmp_path):
    """tests/native/abi_client.c `jac`: G2Projective keys and G1Projective signatures as C structs (random z), through
    blsmi_g2pubs_verify_batch_jac, blsmi_g2_sum_jac / blsmi_g1_sum_jac, blsmi_g2pubs_verify_aggregate_jac, blsmi_g2pubs_sign_batch_jac and
    blsmi_g2_mul_generator_batch_jac (g2pubs/bls.go:159-162, 180-192, 240-270, 132-135, 120-123).  Every printed result is compared with
    the oracle's on the same tuples."""
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "abi_client")
    subprocess.check_call([gcc, "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "abi_client.c"),
                           "-o", exe, "-L", os.path.join(ROOT, "bls_amd"), "-lblsmi", "-Wl,-rpath," + os.path.join(ROOT, "bls_amd")])
    xs = P.XORShift(6021)
    n, m = 11, 7                                                           # the first m tuples are untouched, the rest corrupted in rotation
    msgs, pks, sigs, sks, expect = [], [], [], [], []
    for i in range(n):
        sk = sk_bytes(xs)
        msg = b"jac leg %d" % i + b"x" * (i % 5)
        pk, sig = RC.g2pubs.priv_to_pub(sk), RC.g2pubs.sign(msg, sk)
        good = True
        if i >= m:
            good = False
            kind = i % 3
            if kind == 0:
                pk = RC.g2pubs.priv_to_pub(sk_bytes(xs))                  # another key
            elif kind == 1:
                sig = sig[:48] + ((P.Q - int.from_bytes(sig[48:], "big")) % P.Q).to_bytes(48, "big")     # -signature
            else:
                sig = RC.g2pubs.sign(msg + b"!", sk)                      # a signature over another message
        msgs.append(msg); pks.append(pk); sigs.append(sig); sks.append(sk); expect.append(good)
    assert [RC.g2pubs.verify(a, b, c) for a, b, c in zip(msgs, pks, sigs)] == expect
    jpk = [jac2(xs, p) for p in pks]; jsg = [jac1(xs, s) for s in sigs]   # random Jacobian representatives, Montgomery limbs
    blob = struct.pack("<QQ", n, m)
    for a, b, c, d in zip(msgs, jpk, jsg, sks):
        assert len(b) == 288 and len(c) == 144
        blob += struct.pack("<I", len(a)) + a + b + c + d
    path = tmp_path / "tuples_jac.bin"
    path.write_bytes(blob)
    env = dict(os.environ)
    import torch
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":" + env.get("LD_LIBRARY_PATH", "")   # one HIP runtime per box: torch's copy
    out = subprocess.run([exe, "jac", str(path)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    lines = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l.strip()}
    assert lines["jacbatch"] == ["1" if e else "0" for e in expect]
    # the key sum: a G2Projective with z = 1 whose affine image is the oracle's sum of the affine keys
    assert lines["sumg2inf"] == ["0"]
    s2 = np.array([int(x, 16) for x in lines["sumg2"]], dtype=np.uint64)
    assert RC.g2_jac_to_affine_bytes(s2) == RC.g2_sum(b"".join(pks), n)
    assert P.from_mont(P.from_limbs64(s2[24:30])) == 1 and not any(s2[30:36])          # z = 1 + 0 u
    # aggregates: the oracle's verdicts on the same sums
    agg_m, agg_n = RC.g1_sum(b"".join(sigs[:m]), m), RC.g1_sum(b"".join(sigs), n)
    assert RC.g2pubs.verify_aggregate(agg_m, pks[:m], msgs[:m]) is True and RC.g2pubs.verify_aggregate(agg_n, pks, msgs) is False
    assert lines["aggregate_m"] == ["1"] and lines["aggregate_n"] == ["0"]
    # SignBatch: n G1Projective records, each the oracle's signature
    w = np.array([int(x, 16) for x in lines["signed"]], dtype=np.uint64).reshape(n, 18)
    for i in range(n):
        assert RC.g1_jac_to_affine_bytes(w[i]) == RC.g2pubs.sign(msgs[i], sks[i])
    assert lines["roundtrip"] == ["1"] * n


def test_options_set_before_initialisation_survive_it():
    """blsmi_set_option / blsmi_set_*_threshold called BEFORE the first entry point (or before a re-initialisation after blsmi_shutdown)
    were overwritten by the environment defaults when the library initialised.  Own process: set the row layout off and "lat_rolled" 0 first, then
    run 2 400 pairings -- they must take the one-tuple-per-wave path in its straight-line copy (k_lat:pairing1s), not the row kernels -- and once more
    after shutdown + re-initialisation; BLSMI_ROW_MAX in the environment must NOT win over the explicit call, BLSMI_QUAD_MAX (never set explicitly) must apply."""
    import sys
    code = r'''
import ctypes, numpy as np
from bls_amd import engine as E
from oracle import refcpu as RC
E.set_row_threshold(0, 0); E.set_option("lat_rolled", 0)                    # before any initialisation
lib = E._lib()
n = 2400
g1 = np.frombuffer(RC.g1_generator() * n, dtype=np.uint8); g2 = np.frombuffer(RC.g2_generator() * n, dtype=np.uint8)
def run():
    lib.blsmi_set_profiling(1)
    out = E.pairing_batch(g1, g2, n)
    buf = ctypes.create_string_buffer(8192); lib.blsmi_last_profile(buf, ctypes.c_size_t(8192)); lib.blsmi_set_profiling(0)
    return out, buf.value.decode()
a, p1 = run()
lib.blsmi_shutdown()
b, p2 = run()                                                               # lazily re-initialised
E.set_row_threshold(1, 1 << 20)
c, p3 = run()
assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a[0], RC.pairing_batch(RC.g1_generator(), RC.g2_generator(), 1)[0])
print("PROFILES", p1, "|", p2, "|", p3)
'''
    env = dict(os.environ); env["BLSMI_ROW_MAX"] = "65536"; env["BLSMI_ROW_MIN"] = "1"; env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    p1, p2, p3 = [x.strip() for x in r.stdout.split("PROFILES", 1)[1].split("|")]
    for p in (p1, p2):
        assert "k_lat:pairing1s" in p and "_row" not in p, (p1, p2)
    assert "k_miller1h_row" in p3, p3
