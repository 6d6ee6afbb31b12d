"""-m gpu: the lane-QUAD layout (four lanes per tuple, bls_amd/csrc/quad_body.inc, k_pairing_quad.hip) -- the layout that fills the chip
from 16 384 tuples.  Every Fq12 routine op by op against the oracle, then Pairing on all three paths (one tuple
per wave / lane quad / lane pair) at their boundaries: bit-identical Fq12 (pairing.go:132-136, fq12.go:27-237)."""
import numpy as np
import pytest

from gpu_common import P, RC, pack, rand_fq, rand_g1, rand_g2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from bls_amd import engine
    engine.init(0)
    return engine


def _rand_rec(xs, n, width):
    return np.stack([pack(rand_fq(xs, width)) for _ in range(n)])


def test_fq12_ops_in_lane_quad_layout(eng):
    """odd n: the last quad of a wave and a ragged final workgroup (16 tuples per workgroup) are exercised"""
    xs = P.XORShift(4201)
    n = 37
    a12 = _rand_rec(xs, n, 12); b12 = _rand_rec(xs, n, 12)
    for name, ref in [("FQ12_MUL", lambda x, y: RC.fq12_mul(x, y)), ("FQ12_SQR", lambda x, y: RC.fq12_sqr(x)), ("FQ12_INV", lambda x, y: RC.fq12_inverse(x)[1]),
                      ("FQ12_FROB1", lambda x, y: RC.fq12_frobenius(x, 1)), ("FQ12_FROB2", lambda x, y: RC.fq12_frobenius(x, 2)), ("FQ12_FROB3", lambda x, y: RC.fq12_frobenius(x, 3))]:
        out, _ = eng.debug_op(name, a12, b12 if name == "FQ12_MUL" else None, lane_quad=True)
        assert np.array_equal(out, np.stack([ref(x, y) for x, y in zip(a12, b12)])), name
    # the sparse line multiplication (fq12.go:32-47): (c0, c1, c4) = b[0..5]
    out, _ = eng.debug_op("FQ12_MUL_BY_014", a12, b12, lane_quad=True)
    want = np.stack([RC.fq12_mul_by_014(x, y[0:12], y[12:24], y[24:36]) for x, y in zip(a12, b12)])
    assert np.array_equal(out, want)
    # cyclotomic squarings need subgroup elements: x^((q^6-1)(q^2+1)) of random x, from the oracle
    cyc = []
    for x in a12[:9]:
        inv = RC.fq12_inverse(x)[1]
        conj = x.copy().reshape(12, 6)
        for k in range(6, 12):
            conj[k] = RC.fq_neg(conj[k])
        t = RC.fq12_mul(conj.reshape(-1), inv)
        cyc.append(RC.fq12_mul(RC.fq12_frobenius(t, 2), t))
    cyc = np.stack(cyc)
    out, _ = eng.debug_op("FQ12_CYCLO_SQR", cyc, lane_quad=True)
    assert np.array_equal(out, np.stack([RC.fq12_sqr(x) for x in cyc]))
    out, _ = eng.debug_op("FQ12_CYCLO_RUN16", cyc, lane_quad=True)
    ref = cyc
    for _ in range(16):
        ref = np.stack([RC.fq12_sqr(x) for x in ref])
    assert np.array_equal(out, ref)
    # the unit and zero through the compressed run (z2 = z3 = 0: the alternative fraction of the decompression)
    one = np.zeros((3, 72), dtype=np.uint64); one[:, :6] = pack([1])
    out, _ = eng.debug_op("FQ12_CYCLO_RUN16", one, lane_quad=True)
    assert np.array_equal(out, one)


def test_pairing_on_the_three_paths_agrees_with_the_oracle(eng):
    """the same tuples through one-tuple-per-wave, lane-quad and lane-pair kernels: ragged sizes around the quad's 16-tuple workgroups,
    the reference's generator vector (pairing_test.go:9-58) and points outside the subgroup included"""
    from test_gpu_round3 import _torsion_points
    xs = P.XORShift(4202)
    g1s, g2s = _torsion_points()
    a = [RC.g1_generator()] + [rand_g1(xs) for _ in range(40)] + g1s[:3]
    b = [RC.g2_generator()] + [rand_g2(xs) for _ in range(40)] + g2s[:3]
    n = len(a)
    want = RC.pairing_batch(b"".join(a), b"".join(b), n)
    try:
        for m in (1, 3, 15, 16, 17, n):
            eng.set_latency_threshold(0); eng.set_quad_threshold(1 << 20)          # lane quad
            got = eng.pairing_batch(b"".join(a[:m]), b"".join(b[:m]), m)
            assert np.array_equal(got, want[:m]), ("quad", m)
        eng.set_quad_threshold(0)                                                  # lane pair
        assert np.array_equal(eng.pairing_batch(b"".join(a), b"".join(b), n), want)
        eng.set_latency_threshold(8192)                                            # one tuple per wave
        assert np.array_equal(eng.pairing_batch(b"".join(a), b"".join(b), n), want)
    finally:
        eng.set_latency_threshold(8192); eng.set_quad_threshold(16384)


def test_quad_layout_at_its_design_size(eng):
    """16 384 pairings (1 024 waves: one per SIMD) on the default thresholds take the quad kernels: every row against the lane-pair
    kernels' output, a spread sample against the oracle"""
    n = 16384
    base = 256
    xs = P.XORShift(4203)
    ka = b"".join(P.rand_fr(xs).to_bytes(32, "big") for _ in range(base)); kb = b"".join(P.rand_fr(xs).to_bytes(32, "big") for _ in range(base))
    g1b, _ = eng.g1_mul_generator_batch(ka, base); g2b, _ = eng.g2_mul_generator_batch(kb, base)
    reps = n // base
    g1 = np.ascontiguousarray(np.tile(g1b, (reps, 1)))
    g2 = np.ascontiguousarray(np.concatenate([np.roll(g2b, -r, axis=0) for r in range(reps)]))
    lib = __import__("bls_amd._native", fromlist=["load"]).load()
    import bench
    lib.blsmi_set_profiling(1); bench.read_profile(lib)
    got = eng.pairing_batch(g1.reshape(-1), g2.reshape(-1), n)
    lib.blsmi_set_profiling(0)
    prof = bench.read_profile(lib)
    assert "k_miller1h_quad" in prof and "k_final_exp_quad" in prof, prof
    try:
        eng.set_quad_threshold(0)
        ref = eng.pairing_batch(g1.reshape(-1), g2.reshape(-1), n)
    finally:
        eng.set_quad_threshold(16384)
    bad = np.nonzero((got != ref).any(axis=1))[0]
    assert bad.size == 0, bad[:8]
    for i in (0, 1, 15, 16, 4097, n - 17, n - 1):
        assert np.array_equal(got[i], RC.pairing_batch(g1[i].tobytes(), g2[i].tobytes(), 1)[0]), i


def test_soak_slice_four_layouts_two_limb_representations():
    """a 20-second slice of tools/soak6.py: random batch sizes, special scalars, points outside the subgroup and corrupted verify tuples
    through the wave (15 x 27-bit limbs), row, quad and pair (14 x 28-bit) paths -- same Fq12 bits, same verdicts, samples against the oracle"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["SOAK_SEED"] = "4242"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak6.py"), "20"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "representations agree" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
