"""CPU: the integer model of the endomorphism decompositions (bls_amd/csrc/glv_model.py) against the oracle's Python twin:
the scalar identities, the digit ranges the device ladders rely on, and -- on real curve points -- that the decomposed sums are
the reference's MulFR result (g1.go:80-90, g2.go:92-102), signs included."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bls_amd", "csrc"))
import glv_model as G          # noqa: E402
from oracle import pyref as P  # noqa: E402


def _edge_scalars():
    rnd = random.Random(7)
    return [0, 1, 2, G.Z - 1, G.Z, G.Z + 1, G.Z2 - 1, G.Z2, G.Z2 + 1, G.Z**3 - 1, G.Z**3, P.R_ORDER - 1, P.R_ORDER, P.R_ORDER + 1,
            (1 << 255) - 1, 1 << 255, (1 << 256) - 1] + [rnd.getrandbits(256) for _ in range(3000)] + [rnd.getrandbits(b) for b in (64, 65, 128, 129, 192, 193)]


def test_decompositions_are_exact_and_in_range():
    assert G.Z == P.BLS_X and G.R_ORDER == P.R_ORDER
    for k in _edge_scalars():
        k1, k2 = G.decompose_g1(k)
        assert k1 + k2 * G.Z2 == k and 0 <= k1 < G.Z2 and 0 <= k2 < 1 << 129
        for v in (k1, k2):
            d = G.booth_digits(v, G.G1_WIN, G.G1_NWIN)
            assert len(d) == 26 and all(-16 <= x <= 16 for x in d) and sum(x << (5 * i) for i, x in enumerate(d)) == v
        ds = G.decompose_g2(k)
        assert sum(x * G.Z**i for i, x in enumerate(ds)) == k and all(0 <= x < G.Z for x in ds[:3]) and ds[3] < 1 << 65
        for v in ds:
            d = G.booth_digits(v, G.G2_WIN, G.G2_NWIN)
            assert len(d) == 14 and sum(x << (5 * i) for i, x in enumerate(d)) == v
        r1 = G.lat_record_g1(k); r2 = G.lat_record_g2(k)
        assert (r1 & ((1 << 256) - 1)) < 1 << (4 * G.G1_LAT_NWIN) and (r1 >> 256) < 1 << (4 * G.G1_LAT_NWIN)
        assert all(((r2 >> (128 * i)) & ((1 << 128) - 1)) < 1 << (4 * G.G2_LAT_NWIN) for i in range(4))


def _beta():
    import gen_lat
    return gen_lat.g1_beta()


def test_g1_identity_on_subgroup_points():
    beta = _beta()
    F = P.F1
    xs = P.XORShift(5)
    for _ in range(6):
        a = P.jac_to_affine(F, P.affine_mul(F, P.G1_GEN, P.rand_fr(xs)))
        k = P.rand_int(xs, 1 << 256)
        k1, k2 = G.decompose_g1(k)
        mphi = (beta * a[0] % P.Q, (-a[1]) % P.Q)                           # -phi(P) = (beta x, -y)
        assert P.g1_on_curve(mphi)
        lhs = P.jac_to_affine(F, P.affine_mul(F, a, k))
        rhs = P.jac_to_affine(F, P.jac_add(F, P.affine_mul(F, a, k1), P.affine_mul(F, mphi, k2)))
        assert lhs == rhs
        assert P.jac_to_affine(F, P.affine_mul(F, a, G.Z2)) == mphi       # -phi = [z^2] on the subgroup


def test_g2_identity_on_subgroup_points():
    F = P.F2
    xs = P.XORShift(6)
    for _ in range(4):
        a = P.jac_to_affine(F, P.affine_mul(F, P.G2_GEN, P.rand_fr(xs)))
        k = P.rand_int(xs, 1 << 256)
        d = G.decompose_g2(k)
        p1 = P.psi(a); p2 = P.psi(p1); p3 = P.psi(p2)
        assert P.jac_to_affine(F, P.affine_mul(F, a, G.Z)) == P.affine_neg(F, p1)   # psi = [x] = [-z] on G2
        acc = P.affine_mul(F, a, d[0])
        acc = P.jac_add(F, acc, P.affine_mul(F, P.affine_neg(F, p1), d[1]))
        acc = P.jac_add(F, acc, P.affine_mul(F, p2, d[2]))
        acc = P.jac_add(F, acc, P.affine_mul(F, P.affine_neg(F, p3), d[3]))
        assert P.jac_to_affine(F, acc) == P.jac_to_affine(F, P.affine_mul(F, a, k))
