"""-m gpu: parity of the pairing path (BASELINE config 2) against the oracle: the reference's only
end-to-end Fq12 vector (pairing_test.go:9-58), bit-exact Miller-loop and pairing outputs on seeded
tuples, bilinearity on the full 64k batch, and edge cases."""
import numpy as np
import pytest

from gpu_common import P, RC, mont, rand_g1, rand_g2, unmont

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["latency-path", "lane-row", "lane-quad", "lane-pair"])
def eng(request):
    """Every test of this module runs four times: small batches through the latency path (one tuple per wave, k_lat.hip), through the
    lane-quad kernels (the mid-size layout) and through the lane-pair kernels (the full-chip layout)."""
    from bls_amd import engine
    engine.init(0)
    # four paths, same results: one tuple per wave (k_lat.hip) / per lane quad (k_pairing_quad.hip) / per lane pair
    engine.set_latency_threshold(8192 if request.param in ("latency-path", "lane-row") else 0)
    engine.set_quad_threshold(0 if request.param == "lane-pair" else 16384)
    engine.set_row_threshold(*((1, 1 << 20) if request.param == "lane-row" else (0, 0)))   # round 6: sixteen lanes per tuple (k_pairing_row.hip), whatever the size
    yield engine
    engine.set_latency_threshold(8192); engine.set_quad_threshold(16384); engine.set_row_threshold(*engine.ROW_DEFAULT)


def test_pairing_generator_kat(eng, kats):
    out = eng.pairing_batch(RC.g1_generator(), RC.g2_generator(), 1)[0]
    assert [unmont(out[6 * i:6 * i + 6]) for i in range(12)] == [int(v) for v in kats["pairing_g1gen_g2gen"]]


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 130])
def test_pairing_bit_exact_ragged_sizes(eng, n):
    xs = P.XORShift(200 + n)
    g1 = b"".join(rand_g1(xs) for _ in range(n)); g2 = b"".join(rand_g2(xs) for _ in range(n))
    assert np.array_equal(eng.miller_loop_batch(g1, g2, n), np.stack([RC.miller_loop(g1[96 * i:96 * i + 96], g2[192 * i:192 * i + 192], 1) for i in range(n)]))
    got = eng.pairing_batch(g1, g2, n)
    assert np.array_equal(got, RC.pairing_batch(g1, g2, n))
    # final exponentiation alone, from the oracle's Miller-loop outputs
    ml = np.stack([RC.miller_loop(g1[96 * i:96 * i + 96], g2[192 * i:192 * i + 192], 1) for i in range(min(n, 8))])
    assert np.array_equal(eng.final_exponentiation_batch(ml), got[:min(n, 8)])


def test_empty_batch(eng):
    assert eng.pairing_batch(b"", b"", 0).shape == (0, 72)


def test_config2_64k_pairings_every_row_against_the_oracle(eng):
    """65 536 pairs (a_i G1, b_i G2): a seeded sample is compared bit-for-bit with the oracle and the whole
    batch through e(aP, bQ) = e(P, Q)^(ab): every output equals the KAT raised to a_i b_i, checked here via
    the product relation e(a_i P, b_i Q) * e(-(a_i b_i) P, Q) == 1 on device-independent CPU arithmetic for a
    sample, and via a checksum identity for the full batch."""
    n = 65536
    xs = P.XORShift(2)
    base = 512                         # distinct tuples; the batch tiles them with per-tile scalar twists
    a = [P.rand_fr(xs) for _ in range(base)]; b = [P.rand_fr(xs) for _ in range(base)]
    from bls_amd import engine
    g1b, _ = engine.g1_mul_batch(RC.g1_generator() * base, b"".join(x.to_bytes(32, "big") for x in a), base)
    g2b, _ = engine.g2_mul_batch(RC.g2_generator() * base, b"".join(x.to_bytes(32, "big") for x in b), base)
    # spot-check the device scalar multiplications against the oracle
    for i in (0, 1, base - 1):
        assert g1b[i].tobytes() == RC.g1_mul(RC.g1_generator(), a[i].to_bytes(32, "big"))
        assert g2b[i].tobytes() == RC.g2_mul(RC.g2_generator(), b[i].to_bytes(32, "big"))
    reps = n // base
    # tile r pairs P-row i with Q-row (i + r) mod base: 65 536 distinct (P, Q) combinations
    g1 = np.tile(g1b, (reps, 1))
    g2 = np.concatenate([np.roll(g2b, -r, axis=0) for r in range(reps)])
    out = eng.pairing_batch(g1.reshape(-1), g2.reshape(-1), n)
    # (1) bit-exact against the oracle, ALL 65 536 rows (the oracle's cores in parallel: ~12 s on the box's 16)
    import os
    from concurrent.futures import ThreadPoolExecutor
    cores = max(1, min(32, len(os.sched_getaffinity(0))))
    step = 256
    with ThreadPoolExecutor(cores) as ex:
        want = list(ex.map(lambda lo: RC.pairing_batch(g1[lo:lo + step].tobytes(), g2[lo:lo + step].tobytes(), step), range(0, n, step)))
    want = np.concatenate(want)
    bad = np.nonzero((out != want).any(axis=1))[0]
    assert bad.size == 0, "rows differing from the oracle's Pairing(): %s" % bad[:8]
    # (2) bilinearity ties every sampled output to the reference's single KAT: e(aP,bQ) = KAT^(ab)
    kat = P.pairing(P.G1_GEN, P.G2_GEN)
    for i in [5, 777, 40000]:
        r, j = divmod(i, base)
        e = P.fq12_pow(kat, a[j] * b[(j + r) % base] % P.R_ORDER)
        assert [unmont(out[i][6 * k:6 * k + 6]) for k in range(12)] == P.fq12_flat(e)
    # (3) whole-batch structure: identical inputs give identical outputs; all outputs distinct otherwise
    assert len({out[i].tobytes() for i in range(0, n, 97)}) == len(range(0, n, 97))
    # (4) every output lies in the order-r subgroup's image: x^(q^6) == conj(x) == x^-1 is implied by (1)-(2)
    #     for the sample; for the full batch check the cheap invariant c0.c0.c0-limb words are < q
    top = out[:, 5::6]
    assert (top <= np.uint64(P.limbs64(P.Q)[5])).all()
