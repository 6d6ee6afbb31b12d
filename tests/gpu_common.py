"""Helpers shared by the GPU parity tests: seeded inputs and oracle-side expectations."""
import numpy as np

from oracle import pyref as P
from oracle import refcpu as RC


def mont(v):
    return np.array(P.limbs64(P.to_mont(v)), dtype=np.uint64)


def unmont(l):
    return P.from_mont(P.from_limbs64(l))


def rand_fq(xs, n):
    return [P.rand_int(xs, P.Q) for _ in range(n)]


def pack(vals):
    return np.concatenate([mont(v) for v in vals])


def sk_bytes(xs):
    return P.rand_fr(xs).to_bytes(32, "big")


_G1 = RC.g1_generator()
_G2 = RC.g2_generator()


def rand_g1(xs):
    return RC.g1_mul(_G1, sk_bytes(xs))


def rand_g2(xs):
    return RC.g2_mul(_G2, sk_bytes(xs))


# ---- the reference's in-memory points (bls.G1Projective 18 x u64, bls.G2Projective 36 x u64; the *_jac entry points) ----------
def _f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P.Q, (a[0] * b[1] + a[1] * b[0]) % P.Q)


def g1_to_jac(wire, z=1):
    """96-byte affine wire record (None: infinity) -> the 144-byte in-memory record (x z^2, y z^3, z); z = 0 gives the reference's
    G1ProjectiveZero-shaped (x, y, 0)"""
    if wire is None:
        return np.concatenate([mont(0), mont(1), mont(0)]).tobytes()
    x, y = int.from_bytes(wire[:48], "big"), int.from_bytes(wire[48:], "big")
    return np.concatenate([mont(x * z * z % P.Q), mont(y * z * z * z % P.Q), mont(z % P.Q)]).tobytes()


def g2_to_jac(wire, z=(1, 0)):
    if wire is None:
        return np.concatenate([mont(0), mont(0), mont(1), mont(0), mont(0), mont(0)]).tobytes()
    c = [int.from_bytes(wire[48 * i:48 * i + 48], "big") for i in range(4)]
    z2 = _f2mul(z, z); z3 = _f2mul(z2, z)
    X, Y = _f2mul((c[0], c[1]), z2), _f2mul((c[2], c[3]), z3)
    return np.concatenate([mont(X[0]), mont(X[1]), mont(Y[0]), mont(Y[1]), mont(z[0] % P.Q), mont(z[1] % P.Q)]).tobytes()


def rand_z1(xs):
    return P.rand_int(xs, P.Q - 1) + 1


def rand_z2(xs):
    return (P.rand_int(xs, P.Q - 1) + 1, P.rand_int(xs, P.Q))


def jac1(xs, wire):
    """a random representative (z != 1 with overwhelming probability) of a G1 wire point"""
    return g1_to_jac(wire, rand_z1(xs))


def jac2(xs, wire):
    return g2_to_jac(wire, rand_z2(xs))
