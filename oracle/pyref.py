"""Python big-int twin of the reference's BLS12-381 verify path.  TEST INFRASTRUCTURE ONLY.

This file is part of the *oracle*: it restates, on Python integers in normal (non-Montgomery) form,
the algorithms of phoreproject/bls so that (a) every known-answer test the reference holds for the
hot path can be checked here, (b) golden fixtures under tests/golden/ can be generated, and (c) the C
oracle (oracle/refcpu.c) can be cross-checked by an independent implementation.  Nothing in the
product path (bls_amd/) imports it.

Each function cites the reference file:line it follows (paths relative to /root/reference).
Field elements are ints in [0,q); Fq2 = (c0,c1); Fq6 = (c0,c1,c2) of Fq2; Fq12 = (c0,c1) of Fq6.
Montgomery images (R = 2^384 for Fq, 2^256 for Fr) are produced by to_mont()/limbs helpers only
for comparison with limb-level outputs.
"""
import hashlib
from . import iso_constants as ISO

# ---------------------------------------------------------------------------------------------
# Parameters (fq.go:26, fr.go:16, g2.go:634-636)
# ---------------------------------------------------------------------------------------------
Q = 4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787
R_ORDER = 52435875175126190479447740508185965837690552500527637822603658699938581184513
BLS_X = 0xd201000000010000          # |x|; blsIsNegative = true
RMONT = 1 << 384
RMONT_FR = 1 << 256
MASK64 = (1 << 64) - 1


def to_mont(a):
    return (a * RMONT) % Q


def from_mont(a):
    return (a * pow(RMONT, -1, Q)) % Q


def limbs64(a, n=6):
    return [(a >> (64 * i)) & MASK64 for i in range(n)]


def from_limbs64(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


# ---------------------------------------------------------------------------------------------
# Fq (fq.go)
# ---------------------------------------------------------------------------------------------
Q_MINUS_3_OVER_4 = (Q - 3) // 4        # fq2.go:174
Q_MINUS_1_OVER_2 = (Q - 1) // 2        # fq.go:298


def fq_inv(a):
    """fq.go:224-266 (binary EEA there; value is the modular inverse). 0 -> None."""
    if a % Q == 0:
        return None
    return pow(a, -1, Q)


def fq_sqrt(a):
    """fq.go:203-217: a1 = a^((q-3)/4); a0 = a1^2 a; fail iff a0 == -1; else a1*a."""
    a1 = pow(a, Q_MINUS_3_OVER_4, Q)
    a0 = a1 * a1 % Q * a % Q
    if a0 == Q - 1:
        return None
    return a1 * a % Q


def fq_parity(a):
    """fq.go:269-273: a > -a on normal-form values."""
    return a > (Q - a) % Q


# ---------------------------------------------------------------------------------------------
# Fq2 = Fq[u]/(u^2+1) (fq2.go)
# ---------------------------------------------------------------------------------------------
FQ2_ZERO = (0, 0)
FQ2_ONE = (1, 0)


def fq2_add(a, b):
    return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)


def fq2_sub(a, b):
    return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)


def fq2_neg(a):
    return ((-a[0]) % Q, (-a[1]) % Q)


def fq2_dbl(a):
    return fq2_add(a, a)


def fq2_mul(a, b):
    """fq2.go:116-130."""
    aa = a[0] * b[0]
    bb = a[1] * b[1]
    return ((aa - bb) % Q, ((a[0] + a[1]) * (b[0] + b[1]) - aa - bb) % Q)


def fq2_sqr(a):
    """fq2.go:75-89."""
    return ((a[0] + a[1]) * (a[0] - a[1]) % Q, 2 * a[0] * a[1] % Q)


def fq2_mul_fq(a, s):
    return (a[0] * s % Q, a[1] * s % Q)


def fq2_mul_nr(a):
    """fq2.go:41-45: multiply by 1+u."""
    return ((a[0] - a[1]) % Q, (a[0] + a[1]) % Q)


def fq2_inv(a):
    """fq2.go:133-147."""
    t = fq_inv((a[0] * a[0] + a[1] * a[1]) % Q)
    if t is None:
        return None
    return (a[0] * t % Q, (-a[1] * t) % Q)


def fq2_conj(a):
    return (a[0], (-a[1]) % Q)


def fq2_frob(a, power):
    """fq2.go:149-158: c1 *= coeff[power%2] with coeff = (1, -1)."""
    return a if power % 2 == 0 else fq2_conj(a)


def fq2_cmp(a, b):
    """fq2.go:31-37: compare c1 first, then c0 (normal form)."""
    if a[1] != b[1]:
        return 1 if a[1] > b[1] else -1
    if a[0] != b[0]:
        return 1 if a[0] > b[0] else -1
    return 0


def fq2_parity(a):
    """fq2.go:256-260."""
    return fq2_cmp(a, fq2_neg(a)) > 0


def fq2_pow(a, e):
    r = FQ2_ONE
    for bit in bin(e)[2:] if e else "":
        r = fq2_sqr(r)
        if bit == "1":
            r = fq2_mul(r, a)
    return r


def fq2_sqrt(a):
    """fq2.go:198-232 (Algorithm 9 of eprint 2012/685)."""
    if a == FQ2_ZERO:
        return FQ2_ZERO
    a1 = fq2_pow(a, Q_MINUS_3_OVER_4)
    alpha = fq2_mul(fq2_sqr(a1), a)
    a0 = fq2_mul(fq2_frob(alpha, 1), alpha)
    neg1 = (Q - 1, 0)
    if a0 == neg1:
        return None
    a1 = fq2_mul(a1, a)
    if alpha == neg1:
        return fq2_mul(a1, (0, 1))
    alpha = fq2_add(alpha, FQ2_ONE)
    alpha = fq2_pow(alpha, Q_MINUS_1_OVER_2)
    return fq2_mul(alpha, a1)


# ---------------------------------------------------------------------------------------------
# Fq6 = Fq2[v]/(v^3-(1+u)) (fq6.go)
# ---------------------------------------------------------------------------------------------
FQ6_ZERO = (FQ2_ZERO, FQ2_ZERO, FQ2_ZERO)
FQ6_ONE = (FQ2_ONE, FQ2_ZERO, FQ2_ZERO)


def fq6_add(a, b):
    return tuple(fq2_add(x, y) for x, y in zip(a, b))


def fq6_sub(a, b):
    return tuple(fq2_sub(x, y) for x, y in zip(a, b))


def fq6_neg(a):
    return tuple(fq2_neg(x) for x in a)


def fq6_mul_nr(a):
    """fq6.go:34-37: multiply by v."""
    return (fq2_mul_nr(a[2]), a[0], a[1])


def fq6_mul(a, b):
    """fq6.go:255-292."""
    aa = fq2_mul(a[0], b[0])
    bb = fq2_mul(a[1], b[1])
    cc = fq2_mul(a[2], b[2])
    t1 = fq2_mul(fq2_add(b[1], b[2]), fq2_add(a[1], a[2]))
    t1 = fq2_add(fq2_mul_nr(fq2_sub(fq2_sub(t1, bb), cc)), aa)
    t3 = fq2_mul(fq2_add(b[0], b[2]), fq2_add(a[0], a[2]))
    t3 = fq2_sub(fq2_add(fq2_sub(t3, aa), bb), cc)
    t2 = fq2_mul(fq2_add(b[0], b[1]), fq2_add(a[0], a[1]))
    t2 = fq2_add(fq2_sub(fq2_sub(t2, aa), bb), fq2_mul_nr(cc))
    return (t1, t2, t3)


def fq6_sqr(a):
    """fq6.go:221-252."""
    s0 = fq2_sqr(a[0])
    ab = fq2_mul(a[0], a[1])
    s1 = fq2_dbl(ab)
    s2 = fq2_sqr(fq2_add(fq2_sub(a[0], a[1]), a[2]))
    bc = fq2_mul(a[1], a[2])
    s3 = fq2_dbl(bc)
    s4 = fq2_sqr(a[2])
    c0 = fq2_add(fq2_mul_nr(s3), s0)
    c1 = fq2_add(fq2_mul_nr(s4), s1)
    c2 = fq2_sub(fq2_sub(fq2_add(fq2_add(s1, s2), s3), s0), s4)
    return (c0, c1, c2)


def fq6_mul_by_1(a, c1):
    """fq6.go:40-57."""
    b = fq2_mul(a[1], c1)
    t1 = fq2_mul_nr(fq2_sub(fq2_mul(c1, fq2_add(a[1], a[2])), b))
    t2 = fq2_sub(fq2_mul(c1, fq2_add(a[0], a[1])), b)
    return (t1, t2, b)


def fq6_mul_by_01(a, c0, c1):
    """fq6.go:60-90."""
    aa = fq2_mul(a[0], c0)
    b = fq2_mul(a[1], c1)
    t1 = fq2_add(fq2_mul_nr(fq2_sub(fq2_mul(c1, fq2_add(a[1], a[2])), b)), aa)
    t3 = fq2_add(fq2_sub(fq2_mul(c0, fq2_add(a[0], a[2])), aa), b)
    t2 = fq2_sub(fq2_sub(fq2_mul(fq2_add(c0, c1), fq2_add(a[0], a[1])), aa), b)
    return (t1, t2, t3)


def fq6_inv(a):
    """fq6.go:295-336."""
    c0 = fq2_add(fq2_neg(fq2_mul(fq2_mul_nr(a[2]), a[1])), fq2_sqr(a[0]))
    c1 = fq2_sub(fq2_mul_nr(fq2_sqr(a[2])), fq2_mul(a[0], a[1]))
    c2 = fq2_sub(fq2_sqr(a[1]), fq2_mul(a[0], a[2]))
    tmp = fq2_mul_nr(fq2_add(fq2_mul(a[2], c1), fq2_mul(a[1], c2)))
    tmp = fq2_add(tmp, fq2_mul(a[0], c0))
    tmp = fq2_inv(tmp)
    if tmp is None:
        return None
    return (fq2_mul(tmp, c0), fq2_mul(tmp, c1), fq2_mul(tmp, c2))


def _fq2_pow_int(a, e):
    return fq2_pow(a, e)


# Frobenius coefficients, derived from their definitions (the reference hard-codes Montgomery
# images: fq6.go:144-208, fq12.go:122-168; tests compare a sample of those images).
FROB6_C1 = [_fq2_pow_int((1, 1), (Q ** k - 1) // 3) for k in range(6)]
FROB6_C2 = [_fq2_pow_int((1, 1), (2 * Q ** k - 2) // 3) for k in range(6)]
FROB12_C1 = [_fq2_pow_int((1, 1), (Q ** k - 1) // 6) for k in range(12)]


def fq6_frob(a, power):
    """fq6.go:211-218."""
    return (fq2_frob(a[0], power),
            fq2_mul(fq2_frob(a[1], power), FROB6_C1[power % 6]),
            fq2_mul(fq2_frob(a[2], power), FROB6_C2[power % 6]))


# ---------------------------------------------------------------------------------------------
# Fq12 = Fq6[w]/(w^2-v) (fq12.go)
# ---------------------------------------------------------------------------------------------
FQ12_ONE = (FQ6_ONE, FQ6_ZERO)


def fq12_mul(a, b):
    """fq12.go:198-213."""
    aa = fq6_mul(a[0], b[0])
    bb = fq6_mul(a[1], b[1])
    o = fq6_add(b[0], b[1])
    c1 = fq6_sub(fq6_sub(fq6_mul(fq6_add(a[1], a[0]), o), aa), bb)
    c0 = fq6_add(fq6_mul_nr(bb), aa)
    return (c0, c1)


def fq12_sqr(a):
    """fq12.go:180-195."""
    ab = fq6_mul(a[0], a[1])
    c0c1 = fq6_add(a[0], a[1])
    c0 = fq6_mul(fq6_add(fq6_mul_nr(a[1]), a[0]), c0c1)
    c0 = fq6_sub(c0, ab)
    c1 = fq6_add(ab, ab)
    c0 = fq6_sub(c0, fq6_mul_nr(ab))
    return (c0, c1)


def fq12_conj(a):
    """fq12.go:27-29."""
    return (a[0], fq6_neg(a[1]))


def fq12_mul_by_014(a, c0, c1, c4):
    """fq12.go:32-47."""
    aa = fq6_mul_by_01(a[0], c0, c1)
    bb = fq6_mul_by_1(a[1], c4)
    o = fq2_add(c1, c4)
    r1 = fq6_mul_by_01(fq6_add(a[1], a[0]), c0, o)
    r1 = fq6_sub(fq6_sub(r1, aa), bb)
    r0 = fq6_add(fq6_mul_nr(bb), aa)
    return (r0, r1)


def fq12_inv(a):
    """fq12.go:216-237."""
    t = fq6_sub(fq6_sqr(a[0]), fq6_mul_nr(fq6_sqr(a[1])))
    t = fq6_inv(t)
    if t is None:
        return None
    return (fq6_mul(t, a[0]), fq6_neg(fq6_mul(t, a[1])))


def fq12_frob(a, power):
    """fq12.go:171-177."""
    c0 = fq6_frob(a[0], power)
    c1 = fq6_frob(a[1], power)
    k = FROB12_C1[power % 12]
    return (c0, tuple(fq2_mul(x, k) for x in c1))


def fq12_pow(a, e):
    """fq12.go:108-120 (LSB-first there; value identical)."""
    res = FQ12_ONE
    fi = a
    while e:
        if e & 1:
            res = fq12_mul(res, fi)
        fi = fq12_mul(fi, fi)
        e >>= 1
    return res


def fq12_flat(a):
    """c000,c001,c010,... order of pairing_test.go:9-20."""
    return [a[i][j][k] for i in range(2) for j in range(3) for k in range(2)]


# ---------------------------------------------------------------------------------------------
# G1: y^2 = x^3 + 4 over Fq (g1.go).  Affine = (x,y) or None for infinity; Jacobian = (x,y,z).
# ---------------------------------------------------------------------------------------------
G1_GEN = (3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
          1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569)
B_COEFF = 4
G1_COFACTOR = 76329603384216526031706109802092473003


class _Field:
    """Tiny dispatch so the curve formulas are written once for Fq (G1) and Fq2 (G2)."""

    def __init__(self, add, sub, mul, sqr, neg, inv, zero, one):
        self.add, self.sub, self.mul, self.sqr, self.neg, self.inv, self.zero, self.one = add, sub, mul, sqr, neg, inv, zero, one

    def dbl(self, a):
        return self.add(a, a)


F1 = _Field(lambda a, b: (a + b) % Q, lambda a, b: (a - b) % Q, lambda a, b: a * b % Q, lambda a: a * a % Q,
            lambda a: (-a) % Q, fq_inv, 0, 1)
F2 = _Field(fq2_add, fq2_sub, fq2_mul, fq2_sqr, fq2_neg, fq2_inv, FQ2_ZERO, FQ2_ONE)


def jac_zero(F):
    """g1.go:275 / g2.go:313: (0,1,0)."""
    return (F.zero, F.one, F.zero)


def jac_is_zero(F, p):
    return p[2] == F.zero


def to_jac(F, a):
    return jac_zero(F) if a is None else (a[0], a[1], F.one)


def jac_double(F, p):
    """g1.go:343-397 / g2.go:389-443 (dbl-2009-l)."""
    if jac_is_zero(F, p):
        return p
    x, y, z = p
    a = F.sqr(x)
    b = F.sqr(y)
    c = F.sqr(b)
    d = F.dbl(F.sub(F.sub(F.sqr(F.add(x, b)), a), c))
    e = F.add(F.dbl(a), a)
    f = F.sqr(e)
    nz = F.dbl(F.mul(z, y))
    nx = F.sub(F.sub(f, d), d)
    c8 = F.dbl(F.dbl(F.dbl(c)))
    ny = F.sub(F.mul(F.sub(d, nx), e), c8)
    return (nx, ny, nz)


def jac_add(F, p, o):
    """g1.go:400-482 / g2.go:446-529 (add-2007-bl)."""
    if jac_is_zero(F, p):
        return o
    if jac_is_zero(F, o):
        return p
    z1z1 = F.sqr(p[2])
    z2z2 = F.sqr(o[2])
    u1 = F.mul(p[0], z2z2)
    u2 = F.mul(o[0], z1z1)
    s1 = F.mul(F.mul(p[1], o[2]), z2z2)
    s2 = F.mul(F.mul(o[1], p[2]), z1z1)
    if u1 == u2 and s1 == s2:
        return jac_double(F, p)
    h = F.sub(u2, u1)
    i = F.sqr(F.dbl(h))
    j = F.mul(h, i)
    r = F.dbl(F.sub(s2, s1))
    v = F.mul(u1, i)
    nx = F.sub(F.sub(F.sub(F.sqr(r), j), v), v)
    ny = F.sub(F.mul(F.sub(v, nx), r), F.dbl(F.mul(s1, j)))
    nz = F.mul(F.sub(F.sub(F.sqr(F.add(p[2], o[2])), z1z1), z2z2), h)
    return (nx, ny, nz)


def jac_add_affine(F, p, a):
    """g1.go:485-559 / g2.go:532-606 (madd-2007-bl)."""
    if jac_is_zero(F, p):
        return to_jac(F, a)
    if a is None:
        return p
    z1z1 = F.sqr(p[2])
    u2 = F.mul(a[0], z1z1)
    s2 = F.mul(F.mul(a[1], p[2]), z1z1)
    if p[0] == u2 and p[1] == s2:
        return jac_double(F, p)
    h = F.sub(u2, p[0])
    hh = F.sqr(h)
    i = F.dbl(F.dbl(hh))
    j = F.mul(h, i)
    r = F.dbl(F.sub(s2, p[1]))
    v = F.mul(p[0], i)
    nx = F.sub(F.sub(F.sub(F.sqr(r), j), v), v)
    ny = F.sub(F.mul(F.sub(v, nx), r), F.dbl(F.mul(p[1], j)))
    nz = F.sub(F.sub(F.sqr(F.add(p[2], h)), z1z1), hh)
    return (nx, ny, nz)


def jac_to_affine(F, p):
    """g1.go:322-340 / g2.go:365-386."""
    if jac_is_zero(F, p):
        return None
    zi = F.inv(p[2])
    zi2 = F.sqr(zi)
    return (F.mul(p[0], zi2), F.mul(F.mul(p[1], zi2), zi))


def jac_neg(F, p):
    return (p[0], F.neg(p[1]), p[2])


def affine_mul(F, a, k):
    """g1.go:67-90 / g2.go:79-115: MSB-first double-and-add with mixed addition."""
    res = jac_zero(F)
    for bit in bin(k)[2:] if k else "":
        res = jac_double(F, res)
        if bit == "1":
            res = jac_add_affine(F, res, a)
    return res


def jac_mul(F, p, k):
    """g1.go:562-585 / g2.go:609-632."""
    res = jac_zero(F)
    for bit in bin(k)[2:] if k else "":
        res = jac_double(F, res)
        if bit == "1":
            res = jac_add(F, res, p)
    return res


def affine_neg(F, a):
    return None if a is None else (a[0], F.neg(a[1]))


# G2: y^2 = x^3 + 4(1+u) over Fq2 (g2.go:26-32)
G2_GEN = ((0x24aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
           0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
          (0xce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
           0x606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))
B_COEFF_FQ2 = (4, 4)
G2_COFACTOR = 0x5d543a95414e7f1091d50792876a202cd91de4547085abaa68a205b2e5a7ddfa628f1cb4d9e82ef21537e293a6691ae1616ec6e786f0c70cf1c38e31c7238e5


def g1_on_curve(a):
    return a is None or (a[1] * a[1] - a[0] ** 3 - B_COEFF) % Q == 0


def g2_on_curve(a):
    return a is None or fq2_sqr(a[1]) == fq2_add(fq2_mul(fq2_sqr(a[0]), a[0]), B_COEFF_FQ2)


# ---------------------------------------------------------------------------------------------
# Wire formats (g1.go:157-167,185-249; g2.go:172-186,219-295)
# ---------------------------------------------------------------------------------------------
def fq_bytes(a):
    return a.to_bytes(48, "big")


def g1_serialize(a):
    return fq_bytes(a[0]) + fq_bytes(a[1])


def g2_serialize(a):
    return fq_bytes(a[0][0]) + fq_bytes(a[0][1]) + fq_bytes(a[1][0]) + fq_bytes(a[1][1])


def g1_from_x(x, greatest):
    """g1.go:111-132."""
    y = fq_sqrt((x * x % Q * x + B_COEFF) % Q)
    if y is None:
        return None
    negy = (-y) % Q
    return (x, y if (y < negy) != greatest else negy)


def g2_from_x(x, greatest):
    """g2.go:149-169."""
    y = fq2_sqrt(fq2_add(fq2_mul(fq2_sqr(x), x), B_COEFF_FQ2))
    if y is None:
        return None
    negy = fq2_neg(y)
    return (x, y if (fq2_cmp(y, negy) < 0) != greatest else negy)


def g1_compress(a):
    """g1.go:230-249."""
    if a is None:
        return bytes([0xc0]) + bytes(47)
    b = bytearray(fq_bytes(a[0]))
    if a[1] > (Q - a[1]) % Q:
        b[0] |= 1 << 5
    b[0] |= 1 << 7
    return bytes(b)


def g2_compress(a):
    """g2.go:269-289: x.c1 || x.c0."""
    if a is None:
        return bytes([0xc0]) + bytes(95)
    b = bytearray(fq_bytes(a[0][1]) + fq_bytes(a[0][0]))
    if fq2_cmp(a[1], fq2_neg(a[1])) > 0:
        b[0] |= 1 << 5
    b[0] |= 1 << 7
    return bytes(b)


class DecodeError(Exception):
    pass


def _fq_from_repr_checked(v):
    """FQReprToFQ (fq.go:49-56): values >= q (with a set top nibble) silently become 0."""
    return v if v < Q else 0


def g1_decompress_unchecked(c):
    """g1.go:201-227."""
    c = bytearray(c)
    if c[0] & 0x80 == 0:
        raise DecodeError("unexpected compression mode")
    if c[0] & 0x40:
        c[0] &= 0x3f
        if any(c):
            raise DecodeError("unexpected information in compressed infinity")
        return None
    greatest = bool(c[0] & 0x20)
    c[0] &= 0x1f
    p = g1_from_x(_fq_from_repr_checked(int.from_bytes(c, "big")), greatest)
    if p is None:
        raise DecodeError("point not on curve")
    return p


def g2_decompress_unchecked(c):
    """g2.go:234-267."""
    c = bytearray(c)
    if c[0] & 0x80 == 0:
        raise DecodeError("unexpected compression mode")
    if c[0] & 0x40:
        c[0] &= 0x3f
        if any(c):
            raise DecodeError("unexpected information in infinity point on G2")
        return None
    greatest = bool(c[0] & 0x20)
    c[0] &= 0x1f
    x = (_fq_from_repr_checked(int.from_bytes(c[48:], "big")), _fq_from_repr_checked(int.from_bytes(c[:48], "big")))
    p = g2_from_x(x, greatest)
    if p is None:
        raise DecodeError("point not on curve")
    return p


def g1_in_subgroup(a):
    """g1.go:137-141."""
    return jac_is_zero(F1, affine_mul(F1, a, R_ORDER))


def g2_in_subgroup(a):
    """g2.go:293-295."""
    return jac_is_zero(F2, affine_mul(F2, a, R_ORDER))


def g1_decompress(c):
    p = g1_decompress_unchecked(c)
    if p is not None and not g1_in_subgroup(p):
        raise DecodeError("not in correct subgroup")
    return p


def g2_decompress(c):
    p = g2_decompress_unchecked(c)
    if p is not None and not g2_in_subgroup(p):
        raise DecodeError("point is not in correct subgroup")
    return p


# ---------------------------------------------------------------------------------------------
# G2 prepare + Miller loop + final exponentiation (g2.go:634-801, pairing.go)
# ---------------------------------------------------------------------------------------------
def _doubling_step(r):
    """g2.go:655-708.  r = [x,y,z] mutated in place; returns the line triple."""
    rx, ry, rz = r
    tmp0 = fq2_sqr(rx)
    tmp1 = fq2_sqr(ry)
    tmp2 = fq2_sqr(tmp1)
    tmp3 = fq2_dbl(fq2_sub(fq2_sub(fq2_sqr(fq2_add(tmp1, rx)), tmp0), tmp2))
    tmp4 = fq2_add(fq2_dbl(tmp0), tmp0)
    tmp6 = fq2_add(rx, tmp4)
    tmp5 = fq2_sqr(tmp4)
    zsq = fq2_sqr(rz)
    nx = fq2_sub(fq2_sub(tmp5, tmp3), tmp3)
    nz = fq2_sub(fq2_sub(fq2_sqr(fq2_add(rz, ry)), tmp1), zsq)
    ny = fq2_mul(fq2_sub(tmp3, nx), tmp4)
    tmp2 = fq2_dbl(fq2_dbl(fq2_dbl(tmp2)))
    ny = fq2_sub(ny, tmp2)
    tmp3 = fq2_neg(fq2_dbl(fq2_mul(tmp4, zsq)))
    tmp6 = fq2_sub(fq2_sub(fq2_sqr(tmp6), tmp0), tmp5)
    tmp1 = fq2_dbl(fq2_dbl(tmp1))
    tmp6 = fq2_sub(tmp6, tmp1)
    tmp0 = fq2_dbl(fq2_mul(nz, zsq))
    r[0], r[1], r[2] = nx, ny, nz
    return (tmp0, tmp3, tmp6)


def _addition_step(r, q):
    """g2.go:710-772."""
    rx, ry, rz = r
    zsq = fq2_sqr(rz)
    ysq = fq2_sqr(q[1])
    t0 = fq2_mul(zsq, q[0])
    t1 = fq2_mul(fq2_sub(fq2_sub(fq2_sqr(fq2_add(q[1], rz)), ysq), zsq), zsq)
    t2 = fq2_sub(t0, rx)
    t3 = fq2_sqr(t2)
    t4 = fq2_dbl(fq2_dbl(t3))
    t5 = fq2_mul(t4, t2)
    t6 = fq2_sub(fq2_sub(t1, ry), ry)
    t9 = fq2_mul(t6, q[0])
    t7 = fq2_mul(t4, rx)
    nx = fq2_sub(fq2_sub(fq2_sub(fq2_sqr(t6), t5), t7), t7)
    nz = fq2_sub(fq2_sub(fq2_sqr(fq2_add(rz, t2)), zsq), t3)
    t10 = fq2_add(q[1], nz)
    t8 = fq2_mul(fq2_sub(t7, nx), t6)
    t0 = fq2_dbl(fq2_mul(ry, t5))
    ny = fq2_sub(t8, t0)
    t10 = fq2_sub(fq2_sub(fq2_sqr(t10), ysq), fq2_sqr(nz))
    t9 = fq2_sub(fq2_dbl(t9), t10)
    t10 = fq2_dbl(nz)
    t6 = fq2_dbl(fq2_neg(t6))
    r[0], r[1], r[2] = nx, ny, nz
    return (t10, t6, t9)


X_RSH1 = BLS_X >> 1
# Bits of |x|>>1 below the leading one, MSB first (pairing.go:46-52, g2.go:777-783): 62 entries.
MILLER_BITS = [(X_RSH1 >> i) & 1 for i in range(X_RSH1.bit_length() - 2, -1, -1)]


def g2_prepare(qa):
    """g2.go:650-801: 68 line triples."""
    if qa is None:
        return None
    r = [qa[0], qa[1], FQ2_ONE]
    coeffs = []
    for bit in MILLER_BITS:
        coeffs.append(_doubling_step(r))
        if bit:
            coeffs.append(_addition_step(r, qa))
    coeffs.append(_doubling_step(r))
    return coeffs


def _ell(f, coeffs, p):
    """pairing.go:28-39."""
    c0 = fq2_mul_fq(coeffs[0], p[1])
    c1 = fq2_mul_fq(coeffs[1], p[0])
    return fq12_mul_by_014(f, coeffs[2], c1, c0)


def miller_loop(items):
    """pairing.go:16-75.  items = [(G1 affine, prepared coeffs)], none at infinity."""
    f = FQ12_ONE
    idx = 0
    for bit in MILLER_BITS:
        for p, c in items:
            f = _ell(f, c[idx], p)
        idx += 1
        if bit:
            for p, c in items:
                f = _ell(f, c[idx], p)
            idx += 1
        f = fq12_sqr(f)
    for p, c in items:
        f = _ell(f, c[idx], p)
    return fq12_conj(f)


def _exp_by_x(f, x):
    """pairing.go:92-98."""
    return fq12_conj(fq12_pow(f, x))


def final_exponentiation(r):
    """pairing.go:79-129.  Equals r^(3 (q^12-1)/r_order)."""
    f1 = fq12_conj(r)
    f2 = fq12_inv(r)
    if f2 is None:
        return None
    r = fq12_mul(f1, f2)
    f2 = r
    r = fq12_mul(fq12_frob(r, 2), f2)
    x = BLS_X
    y0 = fq12_sqr(r)
    y1 = _exp_by_x(y0, x)
    y2 = _exp_by_x(y1, x >> 1)
    y3 = fq12_conj(r)
    y1 = fq12_mul(y1, y3)
    y1 = fq12_conj(y1)
    y1 = fq12_mul(y1, y2)
    y2 = _exp_by_x(y1, x)
    y3 = _exp_by_x(y2, x)
    y1 = fq12_conj(y1)
    y3 = fq12_mul(y3, y1)
    y1 = fq12_conj(y1)
    y1 = fq12_frob(y1, 3)
    y2 = fq12_frob(y2, 2)
    y1 = fq12_mul(y1, y2)
    y2 = _exp_by_x(y3, x)
    y2 = fq12_mul(y2, y0)
    y2 = fq12_mul(y2, r)
    y1 = fq12_mul(y1, y2)
    y3 = fq12_frob(y3, 1)
    y1 = fq12_mul(y1, y3)
    return y1


def pairing(p_aff, q_aff):
    """pairing.go:132-136 on affine inputs."""
    return final_exponentiation(miller_loop([(p_aff, g2_prepare(q_aff))]))


def compare_two_pairings(p1, q1, p2, q2):
    """pairing.go:140-147 (affine inputs): e(p1,q1) == e(p2,q2)."""
    f = miller_loop([(p1, g2_prepare(q1)), (affine_neg(F1, p2), g2_prepare(q2))])
    return final_exponentiation(f) == FQ12_ONE


# ---------------------------------------------------------------------------------------------
# Hash to curve (hash.go, g1.go:614-714, g2.go:883-1085)
# ---------------------------------------------------------------------------------------------
def _sha(b):
    return hashlib.sha256(b).digest()


def hp(msg, ctr):
    """hash.go:41-72."""
    mp = _sha(msg) + bytes([ctr])
    t = b"".join(_sha(mp + b"\x01" + bytes([j])) for j in (1, 2))
    return int.from_bytes(t, "big") % Q


def hp2(msg, ctr):
    """hash.go:74-113."""
    mp = _sha(msg) + bytes([ctr])
    out = []
    for i in (1, 2):
        t = b"".join(_sha(mp + bytes([i]) + bytes([j])) for j in (1, 2))
        out.append(int.from_bytes(t, "big") % Q)
    return (out[0], out[1])


def hash_secret_key(b32):
    """hash.go:9-39."""
    mp = _sha(b32) + b"\x00"
    t = b"".join(_sha(mp + b"\x01" + bytes([j])) for j in (1, 2))
    return int.from_bytes(t, "big") % R_ORDER


def _sign_fq(a):
    """g1.go:621-626: -1 iff a > (q-1)/2."""
    return Q - 1 if a > Q_MINUS_1_OVER_2 else 1


def swu_g1_helper(t):
    """g1.go:628-714."""
    A, B = ISO.ellPA, ISO.ellPB
    neg1 = Q - 1
    ndc = neg1 * neg1 % Q
    tsq = t * t % Q
    ndc = (ndc * (tsq * tsq % Q) + neg1 * tsq) % Q
    if ndc == 0:
        x0 = B * fq_inv(neg1 * A % Q) % Q
    else:
        apc = A * ndc % Q
        ndc = (ndc + 1) % Q
        x0 = (-B) % Q * ndc % Q * fq_inv(apc) % Q
    gx0 = (x0 * x0 % Q * x0 + A * x0 + B) % Q
    y = fq_sqrt(gx0)
    if y is not None:
        x = x0
    else:
        x1 = neg1 * tsq % Q * x0 % Q
        gx1 = (x1 * x1 % Q * x1 + A * x1 + B) % Q
        y = fq_sqrt(gx1)
        assert y is not None
        x = x1
    y = y * (_sign_fq(y) * _sign_fq(t) % Q) % Q
    return (x, y)


def _horner(coeffs, x, mul, add):
    v = coeffs[-1]
    for c in reversed(coeffs[:-1]):
        v = add(mul(v, x), c)
    return v


def iso11(p):
    """hash.go:185-206."""
    m = lambda a, b: a * b % Q
    a = lambda a, b: (a + b) % Q
    xn, xd, yn, yd = (_horner(c, p[0], m, a) for c in (ISO.xNum11, ISO.xDen11, ISO.yNum11, ISO.yDen11))
    return (xn * fq_inv(xd) % Q, p[1] * yn % Q * fq_inv(yd) % Q)


def clear_h(p):
    """hash.go:306-309."""
    return jac_to_affine(F1, jac_add_affine(F1, affine_mul(F1, p, BLS_X), p))


def swu_map_g1(t1, t2):
    """hash.go:311-321."""
    pp = swu_g1_helper(t1)
    if t2 is not None:
        pp = jac_to_affine(F1, jac_add_affine(F1, to_jac(F1, pp), swu_g1_helper(t2)))
    return clear_h(iso11(pp))


def hash_g1(msg):
    """hash.go:326-331."""
    m = b"\x01" + msg
    return swu_map_g1(hp(m, 0), hp(m, 1))


def _sign_fq2(f):
    """g2.go:916-931."""
    th = Q_MINUS_1_OVER_2
    if f[1] > th:
        return -1
    if f[1] > 0:
        return 1
    if f[0] > th:
        return -1
    return 1


ELL2P_A = (0, 240)
ELL2P_B = (1012, 1012)
FQ2_NQR = (1, 1)


def swu_g2_helper(t):
    """g2.go:933-1031."""
    ndc = fq2_sqr(FQ2_NQR)
    tsq = fq2_sqr(t)
    t4 = fq2_sqr(tsq)
    ndc = fq2_add(fq2_mul(ndc, t4), fq2_mul(FQ2_NQR, tsq))
    if ndc == FQ2_ZERO:
        x0 = fq2_mul(ELL2P_B, fq2_inv(fq2_mul(FQ2_NQR, ELL2P_A)))
    else:
        apc = fq2_mul(ELL2P_A, ndc)
        ndc = fq2_add(ndc, FQ2_ONE)
        x0 = fq2_mul(fq2_mul(fq2_neg(ELL2P_B), ndc), fq2_inv(apc))
    gx0 = fq2_add(fq2_add(fq2_mul(fq2_sqr(x0), x0), fq2_mul(ELL2P_A, x0)), ELL2P_B)
    s = fq2_sqrt(gx0)
    if s is not None and fq2_sqr(s) == gx0:
        if _sign_fq2(t) != _sign_fq2(s):
            s = fq2_neg(s)
        return (x0, s)
    tcu = fq2_mul(tsq, t)
    t6 = fq2_sqr(tcu)
    x1 = fq2_mul(fq2_mul(FQ2_NQR, tsq), x0)
    gx1 = fq2_mul(fq2_mul(fq2_mul(fq2_sqr(FQ2_NQR), FQ2_NQR), t6), gx0)
    y1 = fq2_sqrt(gx1)
    assert y1 is not None
    if fq2_sqr(y1) == gx1:
        if _sign_fq2(t) != _sign_fq2(y1):
            y1 = fq2_neg(y1)
        return (x1, y1)
    return None


def iso3(p):
    """hash.go:282-303."""
    xn, xd, yn, yd = (_horner(c, p[0], fq2_mul, fq2_add) for c in (ISO.xNum3, ISO.xDen3, ISO.yNum3, ISO.yDen3))
    return (fq2_mul(xn, fq2_inv(xd)), fq2_mul(fq2_mul(p[1], yn), fq2_inv(yd)))


def psi(g):
    """hash.go:341-366."""
    qix = fq2_mul(ISO.iwsc, g[0])
    qix = (qix[0] * ISO.kQiX % Q, (-(qix[1] * ISO.kQiX)) % Q)
    nx = fq2_mul(FQ2_NQR, qix)
    qiy = fq2_mul(ISO.iwsc, g[1])
    qiy = ((qiy[0] + qiy[1]) * ISO.kQiY % Q, (qiy[0] - qiy[1]) * ISO.kQiY % Q)
    ny = fq2_mul(FQ2_NQR, qiy)
    return (nx, ny)


def clear_h2(p):
    """hash.go:368-389."""
    work = affine_mul(F2, p, BLS_X)
    work = jac_add_affine(F2, work, p)
    mpsi = affine_neg(F2, psi(p))
    work = jac_add_affine(F2, work, mpsi)
    work = jac_mul(F2, work, BLS_X)
    work = jac_add_affine(F2, work, mpsi)
    work = jac_add_affine(F2, work, affine_neg(F2, p))
    p2 = jac_to_affine(F2, jac_double(F2, to_jac(F2, p)))
    work = jac_add_affine(F2, work, psi(psi(p2)))
    return jac_to_affine(F2, work)


def swu_map_g2(t1, t2):
    """hash.go:391-402."""
    pp = swu_g2_helper(t1)
    if t2 is not None:
        pp = jac_to_affine(F2, jac_add_affine(F2, to_jac(F2, pp), swu_g2_helper(t2)))
    return clear_h2(iso3(pp))


def hash_g2(msg):
    """hash.go:405-411."""
    m = b"\x01" + msg
    return swu_map_g2(hp2(m, 0), hp2(m, 1))


def hash_g2_with_domain(msg32, domain8):
    """g2.go:1041-1085.  Returns a Jacobian point (ScaleByCofactor result)."""
    x0 = (int.from_bytes(_sha(msg32 + domain8 + b"\x01"), "big"), int.from_bytes(_sha(msg32 + domain8 + b"\x02"), "big"))
    while True:
        gx0 = fq2_add(fq2_mul(fq2_sqr(x0), x0), B_COEFF_FQ2)
        y0 = fq2_sqrt(gx0)
        if y0 is not None:
            if not fq2_parity(y0):
                y0 = fq2_neg(y0)
            return affine_mul(F2, (x0, y0), G2_COFACTOR)
        x0 = fq2_add(x0, FQ2_ONE)


# ---------------------------------------------------------------------------------------------
# Deterministic test streams (g1_test.go:106-124 + crypto/rand.Int semantics)
# ---------------------------------------------------------------------------------------------
class XORShift:
    def __init__(self, state):
        self.state = state

    def read(self, n):
        out = bytearray()
        x = self.state
        for _ in range(n):
            x ^= (x << 13) & MASK64
            x ^= x >> 7
            x ^= (x << 17) & MASK64
            out.append(x & 0xff)
        self.state = x
        return bytes(out)


def rand_int(stream, maxv):
    """Go crypto/rand.Int(reader, max): k=ceil(bitlen(max-1)/8) bytes, top byte masked, retry while >= max."""
    bl = (maxv - 1).bit_length()
    k = (bl + 7) // 8
    b = bl % 8 or 8
    while True:
        by = bytearray(stream.read(k))
        by[0] &= (1 << b) - 1
        v = int.from_bytes(by, "big")
        if v < maxv:
            return v


def rand_fr(stream):
    """fr.go:218-225 (RandFR) -> secret scalar in normal form."""
    return rand_int(stream, R_ORDER)


# ---------------------------------------------------------------------------------------------
# g2pubs / g1pubs (g2pubs/bls.go, g1pubs/bls.go).  Points are affine tuples (None = infinity).
# ---------------------------------------------------------------------------------------------
class G2Pubs:
    """PublicKey in G2, Signature in G1, messages hashed to G1."""

    @staticmethod
    def priv_to_pub(sk):
        return jac_to_affine(F2, affine_mul(F2, G2_GEN, sk))               # g2pubs/bls.go:138-140

    @staticmethod
    def sign(msg, sk):
        return jac_to_affine(F1, affine_mul(F1, hash_g1(msg), sk))         # g2pubs/bls.go:132-135

    @staticmethod
    def verify(msg, pub, sig):
        return compare_two_pairings(sig, G2_GEN, hash_g1(msg), pub)        # g2pubs/bls.go:159-162

    @staticmethod
    def aggregate_sigs(sigs):
        acc = jac_zero(F1)
        for s in sigs:
            acc = jac_add(F1, acc, to_jac(F1, s))
        return jac_to_affine(F1, acc)                                      # g2pubs/bls.go:165-177

    @staticmethod
    def aggregate_pubs(pubs):
        acc = jac_zero(F2)
        for p in pubs:
            acc = jac_add(F2, acc, to_jac(F2, p))
        return jac_to_affine(F2, acc)                                      # g2pubs/bls.go:180-192

    @staticmethod
    def verify_aggregate(sig, pubs, msgs):
        """g2pubs/bls.go:240-270."""
        if len(pubs) != len(msgs):
            return False
        last = b""            # bytes.Equal(m, nil) is true for an empty first message
        for m in sorted(msgs):
            if m == last:
                return False
            last = m
        lhs = pairing(sig, G2_GEN)
        rhs = FQ12_ONE
        for m, pk in zip(msgs, pubs):
            rhs = fq12_mul(rhs, pairing(hash_g1(m), pk))
        return lhs == rhs

    @staticmethod
    def verify_aggregate_common(sig, pubs, msg):
        return G2Pubs.verify(msg, G2Pubs.aggregate_pubs(pubs), sig)         # g2pubs/bls.go:275-278


class G1Pubs:
    """PublicKey in G1, Signature in G2, messages hashed to G2."""

    @staticmethod
    def priv_to_pub(sk):
        return jac_to_affine(F1, affine_mul(F1, G1_GEN, sk))               # g1pubs/bls.go:144-146

    @staticmethod
    def sign(msg, sk):
        return jac_to_affine(F2, affine_mul(F2, hash_g2(msg), sk))         # g1pubs/bls.go:132-135

    @staticmethod
    def sign_with_domain(msg32, sk, domain8):
        h = jac_to_affine(F2, hash_g2_with_domain(msg32, domain8))
        return jac_to_affine(F2, affine_mul(F2, h, sk))                    # g1pubs/bls.go:138-141 (Projective.MulFR; same point)

    @staticmethod
    def verify(msg, pub, sig):
        return compare_two_pairings(G1_GEN, sig, pub, hash_g2(msg))        # g1pubs/bls.go:165-168

    @staticmethod
    def verify_with_domain(msg32, pub, sig, domain8):
        h = jac_to_affine(F2, hash_g2_with_domain(msg32, domain8))
        return compare_two_pairings(G1_GEN, sig, pub, h)                   # g1pubs/bls.go:171-174

    @staticmethod
    def aggregate_sigs(sigs):
        acc = jac_zero(F2)
        for s in sigs:
            acc = jac_add(F2, acc, to_jac(F2, s))
        return jac_to_affine(F2, acc)

    @staticmethod
    def aggregate_pubs(pubs):
        acc = jac_zero(F1)
        for p in pubs:
            acc = jac_add(F1, acc, to_jac(F1, p))
        return jac_to_affine(F1, acc)

    @staticmethod
    def verify_aggregate(sig, pubs, msgs):
        """g1pubs/bls.go:252-282."""
        if len(pubs) != len(msgs):
            return False
        last = b""
        for m in sorted(msgs):
            if m == last:
                return False
            last = m
        lhs = pairing(G1_GEN, sig)
        rhs = FQ12_ONE
        for m, pk in zip(msgs, pubs):
            rhs = fq12_mul(rhs, pairing(pk, hash_g2(m)))
        return lhs == rhs

    @staticmethod
    def verify_aggregate_common(sig, pubs, msg):
        return G1Pubs.verify(msg, G1Pubs.aggregate_pubs(pubs), sig)

    @staticmethod
    def verify_aggregate_with_domain(sig, pubs, msgs32, domain8):
        """g1pubs/bls.go:300-311 (no duplicate-message check)."""
        if len(pubs) != len(msgs32):
            return False
        lhs = pairing(G1_GEN, sig)
        rhs = FQ12_ONE
        for m, pk in zip(msgs32, pubs):
            rhs = fq12_mul(rhs, pairing(pk, jac_to_affine(F2, hash_g2_with_domain(m, domain8))))
        return lhs == rhs
