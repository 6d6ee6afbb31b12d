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
