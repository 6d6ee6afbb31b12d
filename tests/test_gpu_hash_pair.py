"""HashG2 of a large batch runs with a lane pair per message (k_hash_pair.hip: the Fq2 arithmetic in the pairing kernels' layout, the
two maps' exponentiations side by side on the pair).  Same bytes as the one-lane kernel, as the latency path and as the oracle
(hash.go:391-411); messages the pair kernel flags are redone by the one-lane routine."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _msgs(n):
    rng = np.random.default_rng(20260928)
    return [bytes(rng.integers(0, 256, size=int(l), dtype=np.uint8)) for l in rng.integers(0, 150, size=n)]


def _worker():
    """prints the digest of HashG2 over the test messages on the throughput path (environment decides which kernel)"""
    sys.path.insert(0, ROOT)
    from bls_amd import engine
    engine.init(0)
    engine.set_latency_threshold(0)
    print("DIGEST " + hashlib.sha256(engine.hash_g2_batch(_msgs(2051)).tobytes()).hexdigest())
    rng = np.random.default_rng(11)
    print("DOMAIN " + hashlib.sha256(engine.hash_g2_with_domain_batch([rng.bytes(32) for _ in range(1061)], bytes(range(8))).tobytes()).hexdigest())


def _run(env):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l.startswith(("DIGEST ", "DOMAIN "))]


def test_hash_g2_lane_pairs_against_the_latency_path_and_the_oracle():
    from bls_amd import engine
    from gpu_common import RC
    engine.init(0)
    msgs = _msgs(2051) + [b"", b"\x00", b"a" * 55, b"a" * 56, b"a" * 119, b"a" * 120]      # SHA-256 padding boundaries with the 0x01 prefix
    try:
        engine.set_latency_threshold(0); x = engine.hash_g2_batch(msgs)        # lane pairs
        engine.set_latency_threshold(8192); y = engine.hash_g2_batch(msgs)     # SWU lanes + level program
    finally:
        engine.set_latency_threshold(8192)
    assert np.array_equal(x, y)
    for i in list(range(0, len(msgs), 257)) + list(range(len(msgs) - 6, len(msgs))):
        assert x[i].tobytes() == RC.hash_g2(msgs[i]), i
    one = None
    try:
        engine.set_latency_threshold(0); one = engine.hash_g2_batch([b"one"])    # a single pair: the rest of the wave repeats it
    finally:
        engine.set_latency_threshold(8192)
    assert one[0].tobytes() == RC.hash_g2(b"one")


def test_hash_g2_lane_pairs_redo_pass_and_one_lane_kernel_agree():
    a = _run({})                                                                # lane pairs
    b = _run({"BLSMI_HASH_G2_PAIR_REDO_EVERY": "3"})                            # every third message handed to the one-lane routine
    c = _run({"BLSMI_HASH_G2_PAIR": "0"})                                      # the one-lane kernel alone
    assert a == b == c and len(a) == 2
    assert _run({"BLSMI_COFAC2_PAIR": "0"}) == a                                # HashG2WithDomain: the fused one-lane kernel against search kernel + lane-pair ScaleByCofactor


def test_hash_g2_with_domain_shared_search_against_the_latency_path_and_the_oracle():
    """large batches: the 64 lanes of a wave share their messages' try-and-increment search (hash.cuh: hash_g2_with_domain_wave);
    1 061 messages = 16 full waves and a tail of 37, judged in full by the latency path (eight lanes per message + level program)
    and on a sample by the oracle (g2.go:1041-1085)"""
    from bls_amd import engine
    from gpu_common import RC
    engine.init(0)
    rng = np.random.default_rng(7)
    msgs = [rng.bytes(32) for _ in range(1061)]
    dom = bytes(range(1, 9))
    try:
        engine.set_latency_threshold(0); x = engine.hash_g2_with_domain_batch(msgs, dom)
        engine.set_latency_threshold(8192); y = engine.hash_g2_with_domain_batch(msgs, dom)
        engine.set_latency_threshold(0); few = engine.hash_g2_with_domain_batch(msgs[:3], dom)      # a wave with three messages
    finally:
        engine.set_latency_threshold(8192)
    assert np.array_equal(x, y)
    assert np.array_equal(few, x[:3])
    for i in list(range(0, 1061, 97)) + [1024, 1060]:
        assert x[i].tobytes() == RC.hash_g2_with_domain(msgs[i], dom), i


if __name__ == "__main__":
    _worker()
