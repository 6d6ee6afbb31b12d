"""The smallest calls run their square-root exponentiations with ONE field element per wave -- limb j in lane j (fp_row.cuh:
k_swu_g{1,2}_waves, k_tai_g2_waves8, k_g{1,2}_decompress_waves) -- larger ones with one element per lane.  Same outputs: the oracle
judges a sample on either side of every hand-over size, and the two forms must agree on the same inputs."""
import hashlib

import numpy as np
import pytest

from gpu_common import P, RC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from bls_amd import engine
    engine.init(0)
    engine.set_latency_threshold(8192)
    return engine


def _msgs(n, tag):
    return [hashlib.sha256(b"%s-%d" % (tag, i)).digest()[: 1 + i % 32] for i in range(n)]


def test_hash_to_curve_on_both_sides_of_the_wave_hand_over(eng):
    msgs = _msgs(513, b"waves")                                           # 512: one wave per map; 513: one lane per map
    for fn, oracle in ((eng.hash_g1_batch, RC.hash_g1), (eng.hash_g2_batch, RC.hash_g2)):
        small, large = fn(msgs[:512]), fn(msgs)
        assert np.array_equal(small, large[:512])
        for i in (0, 1, 255, 511):
            assert small[i].tobytes() == oracle(msgs[i])
        assert large[512].tobytes() == oracle(msgs[512])
    one = eng.hash_g1_batch([b""])                                         # a single, empty message
    assert one[0].tobytes() == RC.hash_g1(b"")


def test_hash_g2_with_domain_on_both_sides_of_the_wave_hand_over(eng):
    dom = bytes([7, 0, 0, 0, 0, 0, 0, 1])
    m32 = [hashlib.sha256(b"wd-%d" % i).digest() for i in range(129)]     # 128: eight waves per message; 129: eight lanes
    small, large = eng.hash_g2_with_domain_batch(m32[:128], dom), eng.hash_g2_with_domain_batch(m32, dom)
    assert np.array_equal(small, large[:128])
    for i in (0, 5, 127):
        assert small[i].tobytes() == RC.hash_g2_with_domain(m32[i], dom)
    assert large[128].tobytes() == RC.hash_g2_with_domain(m32[128], dom)
    # a message whose first candidates fail: whatever the round count, the eight waves must agree with the eight lanes -- 64 singles
    for i in range(64):
        assert eng.hash_g2_with_domain_batch([m32[i]], dom)[0].tobytes() == large[i].tobytes()


def test_decompression_on_both_sides_of_the_wave_hand_over(eng):
    xs = P.XORShift(77)
    from gpu_common import rand_g1, rand_g2
    g1 = [rand_g1(xs) for _ in range(8)]; g2 = [rand_g2(xs) for _ in range(8)]
    c1 = eng.g1_compress_batch(b"".join(g1), 8); c2 = eng.g2_compress_batch(b"".join(g2), 8)
    n = 513
    in1 = np.tile(c1, (n // 8 + 1, 1))[:n].copy(); in2 = np.tile(c2, (n // 8 + 1, 1))[:n].copy()
    in1[3] = 0; in1[3, 0] = 0xc0                                           # infinity
    in1[5, 47] ^= 1                                                        # most likely not on the curve
    in2[4, 0] &= 0x7f                                                      # compression bit missing
    for fn, data, width in ((eng.g1_decompress_batch, in1, 48), (eng.g2_decompress_batch, in2, 96)):
        for check in (True, False):
            a = fn(data[:512].reshape(-1), 512, check)
            b = fn(data.reshape(-1), n, check)
            for x, y in zip(a, b):
                assert np.array_equal(np.asarray(x), np.asarray(y)[:512])
    out, inf, err = eng.g1_decompress_batch(in1[:8].reshape(-1), 8, True)
    for i in (0, 1, 2, 6, 7):
        assert out[i].tobytes() == g1[i] and not inf[i] and err[i] == 0
    assert inf[3] and err[3] == 0 and err[5] in (3, 4)


def test_hash_to_curve_with_a_row_of_sixteen_lanes_per_map(eng):
    """Between the wave kernels and the lane kernels (513 .. swu_row_max messages) the SWU maps run four per wave, an exponentiation per DPP row
    (k_swu_g{1,2}_rows, fp_row.cuh: fp_pow_row16).  Same points as a lane per map, the oracle judges a sample; ragged counts around the four-row waves."""
    msgs = _msgs(1027, b"rows") + [b""]
    try:
        for fn, oracle in ((eng.hash_g1_batch, RC.hash_g1), (eng.hash_g2_batch, RC.hash_g2)):
            eng.set_option("swu_row_max", 0)
            lanes = fn(msgs)
            eng.set_option("swu_row_max", 4096)
            for n in (513, 514, 515, 1028):
                rows = fn(msgs[-n:])
                assert np.array_equal(rows, lanes[-n:])
            for i in (0, 1, 2, 3, 511, 1026, 1027):
                assert lanes[i].tobytes() == oracle(msgs[i])
        # HashG1 above the four-lane tail's floor (1 280): the other dispatch branch
        big = _msgs(1500, b"rows-big")
        eng.set_option("swu_row_max", 0); lanes = eng.hash_g1_batch(big)
        eng.set_option("swu_row_max", 4096); rows = eng.hash_g1_batch(big)
        assert np.array_equal(rows, lanes)
        for i in (0, 777, 1499):
            assert rows[i].tobytes() == RC.hash_g1(big[i])
    finally:
        eng.set_option("swu_row_max", 4096)
