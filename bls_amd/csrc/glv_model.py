#!/usr/bin/env python3
"""glv_model.py -- integer model of the scalar decompositions behind the endomorphism-accelerated scalar multiplications
(glv.cuh on the device, the 'mul1' / 'mul2' level programs of gen_lat.py, the MSM digit pass).  Everything here is exact
integer arithmetic; tests/test_glv_model.py checks the identities and ranges, the device code restates the same steps on
32-bit words.

The reference multiplies bit-serially (g1.go:80-90, 562-585; g2.go:92-102, 609-632): 255 doublings per 255-bit scalar.  For a
point P of the prime-order subgroup the same group element is reached through the curve's endomorphisms:

  G1: phi(x, y) = (beta x, y) acts as multiplication by lambda = -z^2 (z = |x| = 0xd201000000010000; lambda^2 + lambda + 1 = 0
      mod r).  Euclidean division k = k1 + k2 z^2 (0 <= k1 < z^2 < 2^128, k2 < 2^129 for k < 2^256) gives
          [k] P = [k1] P + [k2] (-phi(P)),                                   -phi(P) = (beta x, -y):
      two half-length scalars on one doubling chain.
  G2: psi (untwist-Frobenius-twist, hash.go:341-366) acts as multiplication by x = -z.  The base-z digits k = sum d_i z^i
      (d_0..d_2 < z < 2^64, d_3 < 2^65 for k < 2^256) give
          [k] P = [d0] P - [d1] psi(P) + [d2] psi^2(P) - [d3] psi^3(P):
      four quarter-length scalars on one doubling chain.

Both are INTEGER identities in k (no reduction mod r anywhere), so any 256-bit scalar is served; they hold for P in the
subgroup -- which is what Sign (the hash point), PrivToPub (the generator) and public keys / signatures that passed
Deserialize's subgroup check are.  For other curve points the library keeps the plain windowed ladder (see blsmi.h).

The divisions are multiplications by precomputed reciprocals: q^ = floor(k M / 2^s) with M = floor(2^s / d), s = 32 (words(k) + 2);
k (2^s/d - M) / 2^s < 2^-64, so q^ is the true quotient or one less, and one conditional correction makes it exact.
"""
Z = 0xd201000000010000
Z2 = Z * Z
R_ORDER = Z**4 - Z**2 + 1
S_BITS = 320                                   # 32 * (8 + 2)
M_Z = (1 << S_BITS) // Z                       # 257 bits, 9 words; for an n-word dividend use M_Z >> (32 (8 - n))
M_Z2 = (1 << S_BITS) // Z2                     # 193 bits, 7 words


def words(v, n):
    assert 0 <= v < 1 << (32 * n)
    return [(v >> (32 * i)) & 0xffffffff for i in range(n)]


def div_recip(k, nwords, d, M320):
    """(q, r) = divmod(k, d) the way the device computes it: reciprocal product, truncation, one correction"""
    assert 0 <= k < 1 << (32 * nwords)
    M = M320 >> (32 * (8 - nwords))
    q = (k * M) >> (32 * (nwords + 2))
    r = k - q * d
    assert 0 <= r < 2 * d, "reciprocal estimate off by more than one"
    if r >= d:
        r -= d; q += 1
    return q, r


def decompose_g1(k):
    """k (any 256-bit integer) -> (k1, k2): k = k1 + k2 z^2, 0 <= k1 < z^2, k2 < 2^129"""
    k2, k1 = div_recip(k, 8, Z2, M_Z2)
    assert k1 + k2 * Z2 == k and k1 < Z2 and k2 < 1 << 129
    return k1, k2


def decompose_g2(k):
    """k -> [d0, d1, d2, d3]: k = sum d_i z^i, d_0..d_2 < z, d_3 < 2^65"""
    q1, d0 = div_recip(k, 8, Z, M_Z)           # q1 < 2^193: 7 words
    q2, d1 = div_recip(q1, 7, Z, M_Z)          # q2 < 2^129: 5 words
    d3, d2 = div_recip(q2, 5, Z, M_Z)          # d3 < 2^65
    assert d0 + Z * (d1 + Z * (d2 + Z * d3)) == k and d3 < 1 << 65
    return [d0, d1, d2, d3]


def booth_digits(v, w, n):
    """n signed base-2^w digits of v by Booth recoding, least significant first, each in [-2^(w-1), 2^(w-1)]:
    digit i = -2^(w-1) b_{wi+w-1} + sum_{j<w-1} 2^j b_{wi+j} + b_{wi-1}  (b_{-1} = 0), read from the w + 1 bits [wi - 1, wi + w)
    of v -- no carry runs between digits, so the ladder can evaluate digit i on the fly from the words of 2 v.
    v = sum dig[i] 2^(w i) as long as bit w n - 1 of v is clear."""
    assert 0 <= v < 1 << (w * n - 1)
    v2 = v << 1
    dig = []
    for i in range(n):
        t = (v2 >> (w * i)) & ((1 << (w + 1)) - 1)
        dig.append((t >> 1) + (t & 1) - ((t >> w) << w))
    assert sum(d << (w * i) for i, d in enumerate(dig)) == v and all(-(1 << (w - 1)) <= d <= 1 << (w - 1) for d in dig)
    return dig


# window shapes of the throughput kernels (glv.cuh): Booth-recoded signed 5-bit digits, table 0 P .. 16 P
G1_WIN, G1_NWIN = 5, 26                        # k1 < 2^128, k2 < 2^129: bit 129 is clear, 26 x 5 = 130
G2_WIN, G2_NWIN = 5, 14                        # digits < 2^65: 14 x 5 = 70
# the level programs (gen_lat.py) use unsigned 4-bit windows of a 512-bit "digit record"
G1_LAT_NWIN = 33                               # 132 bits per half
G2_LAT_NWIN = 17                               # 68 bits per quarter


def lat_record_g1(k):
    """the 64-byte big-endian digit record the G1 level program reads: k1 in bits [0, 256), k2 in bits [256, 512)"""
    k1, k2 = decompose_g1(k)
    return k1 | (k2 << 256)


def lat_record_g2(k):
    d = decompose_g2(k)
    return d[0] | (d[1] << 128) | (d[2] << 256) | (d[3] << 384)


if __name__ == "__main__":
    import random
    rnd = random.Random(1)
    for k in [0, 1, Z - 1, Z, Z2 - 1, Z2, R_ORDER - 1, R_ORDER, (1 << 256) - 1] + [rnd.getrandbits(256) for _ in range(20000)]:
        k1, k2 = decompose_g1(k)
        booth_digits(k1, G1_WIN, G1_NWIN); booth_digits(k2, G1_WIN, G1_NWIN)
        for d in decompose_g2(k):
            booth_digits(d, G2_WIN, G2_NWIN)
        assert lat_record_g1(k) >> 256 < 1 << 132 and (lat_record_g1(k) & ((1 << 256) - 1)) < 1 << 132
    print("glv_model ok")
