#!/usr/bin/env python3
"""gen_lat.py -- programs of the LATENCY path (k_lat.hip): one pairing / one Verify spread across a whole wave.

The throughput kernels give every (message, key, signature) tuple one lane pair, so a call costs the time one lane pair
needs to walk the whole path (~15 ms) however few tuples it carries.  The Go API is one tuple per call
(g2pubs/bls.go:159-162), so small calls get a different decomposition: ONE tuple per 64-lane wave, the tuple's field
elements staged in LDS, and the pairing written as a straight-line program of LEVELS.  In a level every lane executes
the same code on its own job:

    MUL   S[dst] = montmul( sum_i cx_i S[sx_i] , sum_j cy_j S[sy_j] )        one Fq Montgomery product per lane
    LIN   S[dst] = normalise( sum_i c_i S[s_i] )  (value-reduced when asked)   recombination of products
    SQR   S[dst] = 3 ( sum_i c_i S[s_i] )^2                                  the squaring core: levels that hold squarings only
    INV   S[dst] = 1 / S[s]                                                    (one lane; safegcd)
    SEL   S[dst] = S[base + digit_w(record) * stride]                          table entry picked by a 4-bit digit of the tuple's digit record
    LOAD  S[dst] = input record element           OUT / CHECK: results leave LDS

S = LDS slots of one Fq each (15 signed 27-bit limbs, Montgomery R = 2^405, the representation of fp.cuh).  This script
builds the programs: the pairing is written below in a small symbolic tower DSL (Fq values are integer linear
combinations of job results), every product becomes a MUL job, the combinations between products are folded into the
operand gathers or, when they get long / large, materialised by LIN jobs; jobs are levelled as-soon-as-possible into
64-lane levels, LDS slots are allocated by live range.  The limb / value bounds that fp.cuh tracks in its types are
tracked here per linear combination (L = sum |c|, V = sum |c| V_node) and enforced when a job is emitted.

The same file holds an exact big-integer simulator of the level programs (slot reuse included); tests/test_lat_program.py
runs it against the oracle, and the GPU tests run the real kernel against the oracle.

Reference path this replaces for small calls: MillerLoop (pairing.go:16-75), FinalExponentiation (pairing.go:79-129),
CompareTwoPairings (pairing.go:140-147).  The Miller loop here uses homogeneous projective doubling / mixed addition
(depth-2 recurrences instead of the depth-3 Jacobian steps of g2.go:655-772); its value differs from the reference's
Miller value by a factor in Fq2*, which the final exponentiation removes: FinalExponentiation(MillerLoop) -- the only
thing that leaves this path -- is the same field element, bit for bit.
"""
import os
import struct
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
X_ABS = 0xd201000000010000
NLIMB, LB = 15, 27
RMONT = 1 << (NLIMB * LB)

# job / level kinds (shared with k_lat.hip)
K_MUL, K_LIN, K_INV, K_LOAD, K_OUT12, K_CHECK1, K_OUTRAW12, K_OUTAFF, K_ISZERO, K_SEL, K_SQR = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
K_REP = 11            # not a level: "the next `len` levels run `count` times" (rolled squaring runs, see Pairing.exp_by_x)
LANES = 64
TMAX = 7              # terms per MUL operand (descriptor: 7 + 7 term fields)
TLIN = 14             # terms of a LIN job (both operand fields)
LPROD_MAX = 33        # fp.cuh: 15*(La*Lb+1)*2^54 < 2^63
LMAX = 15             # fp.cuh: L*(2^27+64) < 2^31
VPROD_MAX = 1 << 23
V_REDUCE_AT = 512     # a LIN result whose value bound would exceed this is value-reduced (fp_reduce -> V = 3)
CMAX = 15             # |coefficient| of a gathered term


class Node:
    __slots__ = ("id", "kind", "x", "y", "V", "level", "slot", "last", "aux", "reduce", "pin", "ring")

    def __init__(self, nid, kind, x=None, y=None, V=2, aux=None):
        self.id, self.kind, self.x, self.y, self.V, self.aux = nid, kind, x, y, V, aux
        self.level = self.slot = None
        self.last = -1
        self.reduce = False
        self.pin = False
        self.ring = None                 # (run, iteration, slot index): a job of a ROLLED squaring run -- fixed slot, a level of its own (schedule)


class Lin(dict):
    """integer linear combination of nodes: {node: coefficient}"""

    def __add__(self, o):
        r = Lin(self)
        for k, c in o.items():
            v = r.get(k, 0) + c
            if v:
                r[k] = v
            else:
                r.pop(k, None)
        return r

    def __neg__(self):
        return Lin({k: -c for k, c in self.items()})

    def __sub__(self, o):
        return self + (-o)

    def scale(self, s):
        return Lin({k: c * s for k, c in self.items()}) if s else Lin()

    def L(self):
        return sum(abs(c) for c in self.values())

    def V(self):
        return sum(abs(c) * k.V for k, c in self.items())

    def cmax(self):
        return max((abs(c) for c in self.values()), default=0)


class Builder:
    def __init__(self):
        self.nodes = []
        self.consts = {}
        self.inputs = []
        self.out = None

    def _node(self, kind, x=None, y=None, V=2, aux=None):
        n = Node(len(self.nodes), kind, x, y, V, aux)
        self.nodes.append(n)
        return n

    def const(self, v):
        v %= Q
        if v not in self.consts:
            self.consts[v] = self._node("const", V=1, aux=v)
        return Lin({self.consts[v]: 1})

    def table(self, entries):
        """entries: equally long lists of Lin values.  Every value is copied (a LIN job) into a slot of a contiguous, permanently
        reserved block -- entry d, coordinate j at block + d * ncoord + j -- so that a SEL job can index the block with a digit
        of the run-time scalar.  Returns the table id."""
        if not hasattr(self, "tables"):
            self.tables = []
        tid = len(self.tables)
        ncoord = len(entries[0])
        nodes = []
        for d, coords in enumerate(entries):
            assert len(coords) == ncoord
            for j, v in enumerate(coords):
                v = self.flatten(Lin(v))
                assert len(v) <= TLIN and v.L() <= LMAX and v.cmax() <= CMAX
                n = self._node("lin", x=Lin(v), V=max(1, v.V()))
                n.pin = (tid, d * ncoord + j)
                nodes.append(n)
        self.tables.append((ncoord, nodes))
        return tid

    def select(self, tid, window):
        """the ncoord values of table entry digit(window), digit = 4-bit window `window` of the scalar (window 0 = least significant)"""
        ncoord, nodes = self.tables[tid]
        out = []
        for j in range(ncoord):
            deps = Lin({nodes[d * ncoord + j]: 1 for d in range(len(nodes) // ncoord)})       # dependencies, not a sum
            n = self._node("sel", x=deps, V=max(t.V for t in deps), aux=(tid, j, window))
            out.append(Lin({n: 1}))
        return out

    def inp(self, buf, elem, name=""):
        n = self._node("in", V=1, aux=(buf, elem))
        self.inputs.append(n)
        return Lin({n: 1})

    # ---- materialisation ----------------------------------------------------------------------------------------
    def flatten(self, x):
        """inline terms that are themselves un-reduced LIN jobs when the result still fits one gather: a chain of two
        recombination levels becomes one (the inner jobs die if nothing else reads them)"""
        y = Lin()
        changed = False
        for k, c in x.items():
            if k.kind == "lin" and not k.reduce and len(k.x) > 0:
                y = y + k.x.scale(c); changed = True
            else:
                y = y + Lin({k: c})
        if changed and len(y) <= TLIN and y.L() <= LMAX and y.cmax() <= CMAX:
            return y
        return x

    def lin(self, x, force_reduce=False):
        """S = normalise(x): a node again (L = 1)."""
        if not getattr(self, "noflatten", False):
            x = self.flatten(x)
        if len(x) == 1 and not force_reduce:
            (k, c), = x.items()
            if c == 1:
                return Lin(x)
        while len(x) > TLIN or x.L() > LMAX or x.cmax() > CMAX:
            # too long / too large for one gather: materialise a chunk that satisfies the limits, keep the rest
            chunk, rest, l = Lin(), Lin(), 0
            for k, c in sorted(x.items(), key=lambda kc: -abs(kc[1])):
                take = max(-CMAX, min(CMAX, c))
                if len(chunk) < TLIN and l + abs(take) <= LMAX:
                    chunk[k] = take; l += abs(take)
                    if c != take:
                        rest[k] = c - take
                else:
                    rest[k] = c
            assert chunk
            x = rest + self.lin(chunk)
        red = force_reduce or x.V() > V_REDUCE_AT
        n = self._node("lin", x=Lin(x), V=3 if red else x.V())
        n.reduce = red
        return Lin({n: 1})

    def fit(self, x, tmax):
        if len(x) > tmax or x.cmax() > CMAX or x.L() > LMAX:
            return self.lin(x)
        return x

    def mul(self, x, y):
        if not x or not y:
            return Lin()
        x, y = self.fit(x, TMAX), self.fit(y, TMAX)
        while x.L() * y.L() > LPROD_MAX or x.V() * y.V() > VPROD_MAX:
            if x.L() >= y.L() and (len(x) > 1 or x.L() > 1):
                x = self.lin(x, force_reduce=x.V() > 2896)
            elif len(y) > 1 or y.L() > 1:
                y = self.lin(y, force_reduce=y.V() > 2896)
            else:
                x = self.lin(x, force_reduce=True); y = self.lin(y, force_reduce=True)
        return Lin({self._node("mul", x=x, y=y, V=2): 1})

    def sqr3(self, x):
        """3 x^2 as a job of a SQR level: every lane of such a level runs the squaring core (120 + 240 multiply-adds instead of
        225 + 240, one operand gather); the factor 3 rides on the doubled operand (cross terms against 6x, diagonal against 3x),
        which bounds the operand to L <= 3.  Only worth using where a whole level consists of squarings (the cyclotomic
        squaring chains of the final exponentiation)."""
        if not x:
            return Lin()
        x = self.fit(x, TMAX)
        while x.L() > 3 or 3 * x.V() * x.V() > VPROD_MAX:
            x = self.lin(x, force_reduce=x.V() > 1600)
        return Lin({self._node("sqr3", x=x, V=2): 1})

    def inv(self, x):
        x = self.lin(x, force_reduce=True) if (len(x) != 1 or x.L() != 1) else x
        return Lin({self._node("inv", x=x, V=2): 1})


# ---- tower over Lin (Fq2 = (c0, c1), Fq6 = (a0, a1, a2) of Fq2, Fq12 = (c0, c1) of Fq6) ---------------------------
class Tower:
    def __init__(self, b):
        self.b = b
        self.frob_c1 = [self.k2(f2pow((1, 1), (Q**k - 1) // 3)) for k in range(6)]
        self.frob_c2 = [self.k2(f2pow((1, 1), (2 * Q**k - 2) // 3)) for k in range(6)]
        self.frob12_k = [self.k2(f2pow((1, 1), (Q**k - 1) // 6)) for k in range(12)]

    def k2(self, v):
        return (self.b.const(v[0]) if v[0] % Q else Lin(), self.b.const(v[1]) if v[1] % Q else Lin())

    # Fq2
    @staticmethod
    def add2(a, b): return (a[0] + b[0], a[1] + b[1])
    @staticmethod
    def sub2(a, b): return (a[0] - b[0], a[1] - b[1])
    @staticmethod
    def neg2(a): return (-a[0], -a[1])
    @staticmethod
    def sc2(a, s): return (a[0].scale(s), a[1].scale(s))
    @staticmethod
    def nr2(a): return (a[0] - a[1], a[0] + a[1])                   # * (1 + u)   (fq2.go:41-45)
    @staticmethod
    def conj2(a): return (a[0], -a[1])

    def mul2(self, a, b, kar=False):
        m = self.b.mul
        if kar:                                                          # fq2.go:116-130
            p0, p1 = m(a[0], b[0]), m(a[1], b[1])
            p2 = m(a[0] + a[1], b[0] + b[1])
            return (p0 - p1, p2 - p0 - p1)
        return (m(a[0], b[0]) - m(a[1], b[1]), m(a[0], b[1]) + m(a[1], b[0]))

    def sqr2(self, a):                                                   # fq2.go:75-89, the factor 2 rides on an operand
        m = self.b.mul
        return (m(a[0] + a[1], a[0] - a[1]), m(a[0].scale(2), a[1]))

    def mul2_fq(self, a, s):
        return (self.b.mul(a[0], s), self.b.mul(a[1], s))

    def lin2(self, a, force_reduce=False):
        return (self.b.lin(a[0], force_reduce), self.b.lin(a[1], force_reduce))

    # Fq6
    def add6(self, a, b): return tuple(self.add2(x, y) for x, y in zip(a, b))
    def sub6(self, a, b): return tuple(self.sub2(x, y) for x, y in zip(a, b))
    def neg6(self, a): return tuple(self.neg2(x) for x in a)
    def nr6(self, a): return (self.nr2(a[2]), a[0], a[1])               # * v   (fq6.go:34-37)
    def lin6(self, a, fr=False): return tuple(self.lin2(x, fr) for x in a)

    def mul6(self, a, b, kar2=False):                                    # fq6.go:255-292 (Karatsuba over Fq2)
        M = lambda x, y: self.mul2(x, y, kar2)
        aa, bb, cc = M(a[0], b[0]), M(a[1], b[1]), M(a[2], b[2])
        t1 = self.add2(self.nr2(self.sub2(self.sub2(M(self.add2(a[1], a[2]), self.add2(b[1], b[2])), bb), cc)), aa)
        t3 = self.sub2(self.add2(self.sub2(M(self.add2(a[0], a[2]), self.add2(b[0], b[2])), aa), bb), cc)
        t2 = self.add2(self.sub2(self.sub2(M(self.add2(a[0], a[1]), self.add2(b[0], b[1])), aa), bb), self.nr2(cc))
        return (t1, t2, t3)

    def mul6_school(self, a, b):
        """schoolbook over Fq2 (9 products, no operand sums): small L for operands that are sums already"""
        M = self.mul2
        p = [[M(a[i], b[j]) for j in range(3)] for i in range(3)]
        c0 = self.add2(p[0][0], self.nr2(self.add2(p[1][2], p[2][1])))
        c1 = self.add2(self.add2(p[0][1], p[1][0]), self.nr2(p[2][2]))
        c2 = self.add2(self.add2(p[0][2], p[1][1]), p[2][0])
        return (c0, c1, c2)

    def sqr6(self, a):                                                   # fq6.go:221-252
        s0 = self.sqr2(a[0]); s1 = self.sc2(self.mul2(a[0], a[1]), 2)
        s2 = self.sqr2(self.add2(self.sub2(a[0], a[1]), a[2]))
        s3 = self.sc2(self.mul2(a[1], a[2]), 2); s4 = self.sqr2(a[2])
        return (self.add2(self.nr2(s3), s0), self.add2(self.nr2(s4), s1),
                self.sub2(self.sub2(self.add2(self.add2(s1, s2), s3), s0), s4))

    def inv2(self, a):                                                   # fq2.go:133-147
        a = self.lin2(a)
        n = self.b.mul(a[0], a[0]) + self.b.mul(a[1], a[1])
        t = self.b.inv(n)
        return (self.b.mul(a[0], t), -self.b.mul(a[1], t))

    def inv6(self, a):                                                   # fq6.go:295-336
        a = self.lin6(a)
        c0 = self.lin2(self.sub2(self.sqr2(a[0]), self.nr2(self.mul2(a[1], a[2]))))
        c1 = self.lin2(self.sub2(self.nr2(self.sqr2(a[2])), self.mul2(a[0], a[1])))
        c2 = self.lin2(self.sub2(self.sqr2(a[1]), self.mul2(a[0], a[2])))
        t = self.add2(self.nr2(self.add2(self.mul2(a[2], c1), self.mul2(a[1], c2))), self.mul2(a[0], c0))
        ti = self.lin2(self.inv2(t))
        return (self.mul2(ti, c0), self.mul2(ti, c1), self.mul2(ti, c2))

    # Fq12
    def lin12(self, a, fr=False): return (self.lin6(a[0], fr), self.lin6(a[1], fr))
    def conj12(self, a): return (a[0], self.neg6(a[1]))                 # fq12.go:27-29

    def mul12(self, a, b):                                               # fq12.go:198-213
        sa, sb = self.lin6(self.add6(a[0], a[1])), self.lin6(self.add6(b[0], b[1]))   # sums first: all 54 products then share one level
        aa = self.mul6(a[0], b[0], kar2=True)
        bb = self.mul6(a[1], b[1], kar2=True)
        t = self.mul6(sa, sb, kar2=True)
        return self.lin12((self.add6(self.nr6(bb), aa), self.sub6(self.sub6(t, aa), bb)))

    def sqr12(self, a):                                                  # fq12.go:180-195
        s1 = self.lin6(self.add6(self.nr6(a[1]), a[0])); s2 = self.lin6(self.add6(a[0], a[1]))
        ab = self.mul6(a[0], a[1], kar2=True)
        t = self.mul6(s1, s2, kar2=True)
        return self.lin12((self.sub6(self.sub6(t, ab), self.nr6(ab)), self.add6(ab, ab)))

    def inv12(self, a):                                                  # fq12.go:216-237
        a = self.lin12(a)
        t = self.inv6(self.lin6(self.sub6(self.sqr6(a[0]), self.nr6(self.lin6(self.sqr6(a[1]))))))
        t = self.lin6(t)
        return self.lin12((self.mul6(t, a[0]), self.neg6(self.mul6(t, a[1]))))

    def frob12(self, a, p):                                              # fq12.go:171-177, fq6.go:211-218, fq2.go:156-158
        fr2 = (lambda z: self.conj2(z)) if p % 2 else (lambda z: z)
        def frob6(c):
            return (fr2(c[0]), self.mul2(fr2(c[1]), self.frob_c1[p % 6]), self.mul2(fr2(c[2]), self.frob_c2[p % 6]))
        c0 = frob6(a[0]); c1 = self.lin6(frob6(a[1]))
        k = self.frob12_c(p)
        return self.lin12((c0, tuple(self.mul2(z, k) for z in c1)))

    def frob12_c(self, p):
        return self.frob12_k[p % 12]

    def cyc_sqr(self, f, ring=None):
        """Granger-Scott squaring on the cyclotomic subgroup (same element as fq12.go:180-195 there), written with SQUARINGS only,
        so that its product level runs the squaring core (SQR level).  For each Fq4 pair (a, b), with S(x) = 3 x^2:
            3 (a^2 + xi b^2) = (S(a0) - S(a1) + 2 S(b0) - S(b0+b1),   S(a0+a1) - S(a0) - S(a1) + S(b0+b1) - 2 S(b1))
            3 (2ab)          = (S(a0+b0) - S(a0) - S(b0) - S(a1+b1) + S(a1) + S(b1),   S(a0+b1) + S(a1+b0) - S(a0) - S(b1) - S(a1) - S(b0))
        10 squarings of operands with L <= 2 per pair, 30 per squaring; every output is ONE linear job of <= 8 terms, L <= 10."""
        S = self.b.sqr3
        first_node = len(self.b.nodes)
        if ring is not None:
            self.b.noflatten = True                                        # every output is exactly ONE job over this iteration's squares and the previous outputs
        z0, z4, z3 = f[0]; z2, z1, z5 = f[1]
        def fp4(a, b):
            sa0, sa1, sb0, sb1 = S(a[0]), S(a[1]), S(b[0]), S(b[1])
            sa, sb = S(a[0] + a[1]), S(b[0] + b[1])
            s00, s11, s01, s10 = S(a[0] + b[0]), S(a[1] + b[1]), S(a[0] + b[1]), S(a[1] + b[0])
            A = (sa0 - sa1 + sb0.scale(2) - sb, sa - sa0 - sa1 + sb - sb1.scale(2))
            B = (s00 - sa0 - sb0 - s11 + sa1 + sb1, s01 + s10 - sa0 - sb1 - sa1 - sb0)
            return A, B                                                    # 3 (a^2 + xi b^2), 3 (2ab)
        a0, a1 = fp4(z0, z1); b0, b1 = fp4(z2, z3); c0, c1 = fp4(z4, z5)
        L2 = lambda t: self.lin2(t, bool(ring is not None and ring[2]))
        r00 = L2(self.sub2(a0, self.sc2(z0, 2)))
        r11 = L2(self.add2(a1, self.sc2(z1, 2)))                          # 3 * 2ab + 2 z1
        r01 = L2(self.sub2(b0, self.sc2(z4, 2)))
        r12 = L2(self.add2(b1, self.sc2(z5, 2)))
        r10 = L2(self.add2(self.nr2(c1), self.sc2(z2, 2)))                # 3 xi (2ab) + 2 z2
        r02 = L2(self.sub2(c0, self.sc2(z3, 2)))
        if ring is not None:
            # a ROLLED iteration: its 30 squarings and 12 recombinations get fixed slots -- squares: ring 0..29 (dead after this iteration's
            # recombination level), outputs: ring 30 + 12 (iteration mod 2) + j (ping-pong: iteration i + 1 reads them while it writes its own)
            self.b.noflatten = False
            run, it, _ = ring
            outs = [c for pair in ((r00, r01, r02), (r10, r11, r12)) for q in pair for c in q]
            out_nodes = []
            for c in outs:
                assert len(c) == 1 and list(c.values()) == [1]
                out_nodes.append(next(iter(c)))
            new = self.b.nodes[first_node:]
            sq = [n for n in new if n.kind == "sqr3"]
            assert len(sq) == 30 and len(out_nodes) == 12 and all(n.kind == "lin" for n in out_nodes)
            for k, n in enumerate(sq):
                n.ring = (run, it, k)
            for j, n in enumerate(out_nodes):
                n.ring = (run, it, 30 + 12 * (it & 1) + j)
        return ((r00, r01, r02), (r10, r11, r12))

    def mul12_sparse5(self, f, l0, l1, l2, l4, l5):
        """f * (l0 + l1 v + l2 v^2 + (l4 v + l5 v^2) w)  (the product of two line values, tower_body.inc:fp12_mul_by_line_pair)"""
        m0 = (l0, l1, l2)
        m01 = self.lin6((l0, self.add2(l1, l4), self.add2(l2, l5)))
        fs = self.lin6(self.add6(f[0], f[1]))
        t0 = self.mul6(f[0], m0, kar2=True)
        # (a0 + a1 v + a2 v^2)(b1 v + b2 v^2)
        a = f[1]
        p11, p22 = self.mul2(a[1], l4, True), self.mul2(a[2], l5, True)
        cross = self.sub2(self.sub2(self.mul2(self.add2(a[1], a[2]), self.add2(l4, l5), True), p11), p22)
        t1 = (self.nr2(cross), self.add2(self.mul2(a[0], l4, True), self.nr2(p22)), self.add2(self.mul2(a[0], l5, True), p11))
        t2 = self.mul6(fs, m01, kar2=True)
        return self.lin12((self.add6(self.nr6(t1), t0), self.sub6(self.sub6(t2, t0), t1)))

    def mul12_by_014(self, f, c0, c1, c4):                               # fq12.go:32-47
        def by01(a, k0, k1):
            aa, bb = self.mul2(a[0], k0, True), self.mul2(a[1], k1, True)
            t1 = self.add2(self.nr2(self.sub2(self.mul2(k1, self.add2(a[1], a[2]), True), bb)), aa)
            t3 = self.add2(self.sub2(self.mul2(k0, self.add2(a[0], a[2]), True), aa), bb)
            t2 = self.sub2(self.sub2(self.mul2(self.add2(k0, k1), self.add2(a[0], a[1]), True), aa), bb)
            return (t1, t2, t3)
        def by1(a, k1):
            bb = self.mul2(a[1], k1, True)
            t1 = self.nr2(self.sub2(self.mul2(k1, self.add2(a[1], a[2]), True), bb))
            t2 = self.sub2(self.mul2(k1, self.add2(a[0], a[1]), True), bb)
            return (t1, t2, bb)
        fs, c14 = self.lin6(self.add6(f[1], f[0])), self.lin2(self.add2(c1, c4))
        aa = by01(f[0], c0, c1); bb = by1(f[1], c4)
        t = by01(fs, c0, c14)
        return self.lin12((self.add6(self.nr6(bb), aa), self.sub6(self.sub6(t, aa), bb)))

    def line_pair(self, a, b):
        """product of two line values a0 + a1 v + a4 v w (given as (c0, c1, c4)): five coefficients l0, l1, l2, l4, l5"""
        M = lambda x, y: self.mul2(x, y, True)
        p00, p11, p44 = M(a[0], b[0]), M(a[1], b[1]), M(a[2], b[2])
        l0 = self.add2(p00, self.nr2(p44))
        l1 = self.sub2(self.sub2(M(self.add2(a[0], a[1]), self.add2(b[0], b[1])), p00), p11)
        l4 = self.sub2(self.sub2(M(self.add2(a[0], a[2]), self.add2(b[0], b[2])), p00), p44)
        l5 = self.sub2(self.sub2(M(self.add2(a[1], a[2]), self.add2(b[1], b[2])), p11), p44)
        return tuple(self.lin2(z) for z in (l0, l1, p11, l4, l5))


def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2pow(a, e):
    r = (1, 0)
    while e:
        if e & 1: r = f2mul(r, a)
        a = f2mul(a, a); e >>= 1
    return r


# ---- pairing programs ------------------------------------------------------------------------------------------------
B_TWIST = (4, 4)                       # E': y^2 = x^3 + 4(1 + u)   (g2.go:12-16)


class Pairing:
    def __init__(self, b):
        self.b = b
        self.T = Tower(b)
        self.b3 = self.T.k2(f2mul((3, 0), B_TWIST))          # 3 b'
        self.rolled = True                                   # exp_by_x: squaring runs as K_REP loops (build_program says when not)

    def dbl_step(self, R):
        """homogeneous projective doubling on E' (y^2 = x^3 + b', b' = 4 xi) and the tangent line at R, as (c0, o1, o0) of the
        line value c0 + (o1 xP) v + (o0 yP) v w (the 0/1/4 coefficients of pairing.go:28-39):
            A = XY, B = Y^2, E = 3 b' Z^2 = 12 xi Z^2, H = 2YZ, F = 3E, G = B + F
            X3 = 2 A (B - F),  Y3 = G^2 - 12 E^2,  Z3 = 4 B H          line: (B - E) - 3 X^2 xP v + H yP v w
        Two product levels deep.  E comes straight out of the first level: 12 (z0+z1)(z0-z1) and 24 z0 z1 are products of
        scaled operands, (z0+z1) and (z0-z1) having been materialised together with Z."""
        T, m = self.T, self.b.mul
        X, Y, Z, zs, zd = R
        A = T.mul2(X, Y)
        Bq = T.sqr2(Y); X2 = T.sqr2(X)
        H = T.sc2(T.mul2(Y, Z), 2)
        p12 = m(zs.scale(4), zd.scale(3)); r24 = m(Z[0].scale(4), Z[1].scale(6))
        E = (p12 - r24, p12 + r24)                                    # 12 xi Z^2
        F = T.sc2(E, 3)
        G = T.add2(Bq, F)
        gs, gd = self.b.lin(G[0] + G[1]), self.b.lin(G[0] - G[1])
        Gl = T.lin2(G)
        X3 = T.mul2(T.sc2(A, 2), T.sub2(Bq, F))
        G2 = (m(gs, gd), m(Gl[0].scale(2), Gl[1]))
        Y3 = T.sub2(G2, T.sc2(T.sqr2(E), 12))
        Z3 = T.mul2(T.sc2(Bq, 4), H)
        Z3l = T.lin2(Z3)
        line = (T.lin2(T.sub2(Bq, E)), T.sc2(X2, -3), H)
        return (T.lin2(X3), T.lin2(Y3), Z3l, self.b.lin(Z3[0] + Z3[1]), self.b.lin(Z3[0] - Z3[1])), line

    def add_step(self, R, Qa):
        """mixed addition R + Q (Q affine) and the chord through R and Q"""
        T = self.T
        X, Y, Z = R[0], R[1], R[2]
        xq, yq = Qa
        th = T.lin2(T.sub2(Y, T.mul2(yq, Z))); la = T.lin2(T.sub2(X, T.mul2(xq, Z)))
        C = T.lin2(T.sqr2(th)); D = T.lin2(T.sqr2(la))
        E = T.lin2(T.mul2(la, D)); F = T.lin2(T.mul2(Z, C)); G = T.lin2(T.mul2(X, D))
        Hh = T.lin2(T.sub2(T.add2(E, F), T.sc2(G, 2)))
        X3 = T.mul2(la, Hh)
        Y3 = T.sub2(T.mul2(th, T.lin2(T.sub2(G, Hh))), T.mul2(E, Y))
        Z3 = T.mul2(Z, E)
        line = (T.lin2(T.sub2(T.mul2(th, xq), T.mul2(la, yq))), T.neg2(th), la)
        return (T.lin2(X3), T.lin2(Y3), T.lin2(Z3), self.b.lin(Z3[0] + Z3[1]), self.b.lin(Z3[0] - Z3[1])), line

    def dbl_step_ref(self, R):
        """doubling step with the reference's formulas (g2.go:655-708): the line coefficients are the reference's field
        elements (G2Prepared), which the Miller VALUE -- not only the pairing -- depends on.  Three product levels."""
        T = self.T
        X, Y, Z = R[0], R[1], R[2]
        tmp0, tmp1, zsq = T.lin2(T.sqr2(X)), T.lin2(T.sqr2(Y)), T.lin2(T.sqr2(Z))
        nz = T.lin2(T.sub2(T.sub2(T.sqr2(T.add2(Z, Y)), tmp1), zsq))
        tmp2 = T.lin2(T.sqr2(tmp1))
        tmp3 = T.lin2(T.sc2(T.sub2(T.sub2(T.sqr2(T.add2(tmp1, X)), tmp0), tmp2), 2))
        tmp4 = T.sc2(tmp0, 3)
        tmp5 = T.lin2(T.sqr2(tmp4))
        tmp6 = T.add2(X, tmp4)
        nx = T.lin2(T.sub2(tmp5, T.sc2(tmp3, 2)))
        ny = T.lin2(T.sub2(T.mul2(T.sub2(tmp3, nx), tmp4, kar=True), T.sc2(tmp2, 8)))
        o1 = T.sc2(T.mul2(tmp4, zsq, kar=True), -2)
        o2 = T.lin2(T.sub2(T.sub2(T.sub2(T.sqr2(tmp6), tmp0), tmp5), T.sc2(tmp1, 4)))
        o0 = T.sc2(T.mul2(nz, zsq, kar=True), 2)
        return (nx, ny, nz), (o2, o1, o0)

    def add_step_ref(self, R, Qa, ysq):
        """addition step with the reference's formulas (g2.go:710-772)"""
        T = self.T
        X, Y, Z = R[0], R[1], R[2]
        qx, qy = Qa
        zsq = T.lin2(T.sqr2(Z))
        t0 = T.mul2(zsq, qx, kar=True)
        t1 = T.lin2(T.mul2(T.lin2(T.sub2(T.sub2(T.sqr2(T.add2(qy, Z)), ysq), zsq)), zsq, kar=True))
        t2 = T.lin2(T.sub2(t0, X))
        t3 = T.lin2(T.sqr2(t2))
        t4 = T.sc2(t3, 4)
        t5 = T.lin2(T.mul2(t4, t2, kar=True))
        t6 = T.lin2(T.sub2(t1, T.sc2(Y, 2)))
        t9 = T.mul2(t6, qx, kar=True)
        t7 = T.lin2(T.mul2(t4, X, kar=True))
        nx = T.lin2(T.sub2(T.sub2(T.sqr2(t6), t5), T.sc2(t7, 2)))
        nz = T.lin2(T.sub2(T.sub2(T.sqr2(T.add2(Z, t2)), zsq), t3))
        t8 = T.mul2(T.sub2(t7, nx), t6, kar=True)
        ny = T.lin2(T.sub2(t8, T.sc2(T.mul2(Y, t5, kar=True), 2)))
        t10 = T.sub2(T.sub2(T.sqr2(T.add2(qy, nz)), ysq), T.sqr2(nz))
        o2 = T.lin2(T.sub2(T.sc2(t9, 2), t10))
        return (nx, ny, nz), (o2, T.sc2(t6, -2), T.sc2(nz, 2))

    def eval_line(self, line, P):
        """(c0, o1, o0) at P = (xP, yP): the 014 element (c0, o1 xP, o0 yP)"""
        T = self.T
        return (line[0], T.lin2(T.mul2_fq(line[1], P[0])), T.lin2(T.mul2_fq(line[2], P[1])))

    def miller(self, pairs, exact=False):
        """prod_k MillerLoop(P_k, Q_k) for 1 or 2 pairs: g <- g^2 * lines, lines of a step multiplied together first.
        exact: the reference's doubling / addition steps -- the result is the reference's Miller value itself (pairing.go:16-75)"""
        T = self.T
        one = self.b.const(1)
        Rs = [(q[0], q[1], (one, Lin()), one, one) for _, q in pairs]            # (X, Y, Z, z0 + z1, z0 - z1) with Z = 1
        ysq = [T.lin2(T.sqr2(q[1])) for _, q in pairs] if exact else None
        g = None
        xr = X_ABS >> 1

        def absorb(g, lines, square):
            if len(lines) == 2:
                m = T.line_pair(lines[0], lines[1])
                if g is None:
                    z = (Lin(), Lin())
                    return ((m[0], m[1], m[2]), (z, m[3], m[4]))
                if square:
                    g = T.sqr12(g)
                return T.mul12_sparse5(g, *m)
            l = lines[0]
            if g is None:
                z = (Lin(), Lin())
                return ((l[0], l[1], z), (z, l[2], z))
            if square:
                g = T.sqr12(g)
            return T.mul12_by_014(g, l[0], l[1], l[2])

        steps = []
        for i in range(61, -1, -1):
            steps.append("dbl")
            if (xr >> i) & 1:
                steps.append("add")
        steps.append("dbl")
        # g_k = g_{k-1}^2 * l_k for doubling steps (the reference's f <- (f l)^2 shifted by one, same sequence of values up to
        # the last squaring it does not do either), g_k = g_{k-1} * l_k for addition steps
        for s in steps:
            lines = []
            for k, (P, Qa) in enumerate(pairs):
                if exact:
                    Rs[k], ln = (self.dbl_step_ref(Rs[k]) if s == "dbl" else self.add_step_ref(Rs[k], Qa, ysq[k]))
                else:
                    Rs[k], ln = (self.dbl_step(Rs[k]) if s == "dbl" else self.add_step(Rs[k], Qa))
                lines.append(self.eval_line(ln, P))
            g = absorb(g, lines, square=(s == "dbl"))
        return T.conj12(g)                                          # x < 0 (pairing.go:71-73)

    def exp_by_x(self, f, e):                                          # pairing.go:92-98
        """ROLLED (self.rolled, programs '...r'): a run of n >= 4 squarings between two multiplications -- the zero runs of |x| -- becomes
        iteration 0, then a LOOP over iterations (1, 2), (3, 4), ... (K_REP: the kernel runs the same four levels (n - 2) // 2 times), then what
        is left, straight-line.  For the levels to repeat byte for byte the iterations use fixed slots (ping-pong, Tower.cyc_sqr) and a fixed
        value-reduction pattern: every EVEN iteration reduces its outputs (the straight-line form reduces when the bound demands it, about
        every fifth squaring).  The last iteration of a run is an ordinary squaring: its outputs live on after the run."""
        T = self.T
        res = f
        bits = [(e >> i) & 1 for i in range(e.bit_length() - 2, -1, -1)]
        k = 0
        while k < len(bits):
            run = 1
            while bits[k + run - 1] == 0 and k + run < len(bits):
                run += 1                                               # squarings up to and including the one whose bit is set (or the last)
            if self.rolled and run >= 4:
                rid = self.nruns = getattr(self, "nruns", 0) + 1
                for it in range(run - 1):
                    res = T.cyc_sqr(res, ring=(rid, it, it % 2 == 0))
                res = T.cyc_sqr(res)
            else:
                for _ in range(run):
                    res = T.cyc_sqr(res)
            k += run
            if bits[k - 1]:
                res = T.mul12(res, f)
        return T.conj12(res)

    def final_exp(self, r):                                            # pairing.go:79-129 (same chain, same power 3 (q^12 - 1) / r)
        T = self.T
        r = T.lin12(r, True)
        f2 = T.inv12(r)
        r = T.mul12(T.conj12(r), f2)
        f2 = T.frob12(r, 2)
        r = T.mul12(f2, r)
        x = X_ABS
        y0 = T.cyc_sqr(r)
        y1 = self.exp_by_x(y0, x)
        y2 = self.exp_by_x(y1, x >> 1)
        y3 = T.conj12(r)
        y1 = T.mul12(y1, y3)
        y1 = T.conj12(y1)
        y1 = T.mul12(y1, y2)
        y2 = self.exp_by_x(y1, x)
        y3 = self.exp_by_x(y2, x)
        y1 = T.conj12(y1)
        y3 = T.mul12(y3, y1)
        y1 = T.conj12(y1)
        y1 = T.frob12(y1, 3)
        y2 = T.frob12(y2, 2)
        y1 = T.mul12(y1, y2)
        y2 = self.exp_by_x(y3, x)
        y2 = T.mul12(y2, y0)
        y2 = T.mul12(y2, r)
        y1 = T.mul12(y1, y2)
        y3 = T.frob12(y3, 1)
        return T.mul12(y1, y3)


# ---- curve arithmetic for the hash-to-curve tails (hash.go:185-389, g2.go:104-138) --------------------------------------
# Points are homogeneous projective (X : Y : Z) on y^2 = x^3 + b and every group operation uses the COMPLETE formulas of
# Renes-Costello-Batina (a = 0): doubling, addition of equal / opposite points and the point at infinity (0 : 1 : 0) need no
# case distinction, and both operations are two product levels deep (the Jacobian formulas of g1.go / g2.go are five).
import importlib.util as _ilu
import pathlib as _pl
_spec = _ilu.spec_from_file_location("iso_data", _pl.Path(__file__).resolve().parent / "iso_data.py")
ISO = _ilu.module_from_spec(_spec); _spec.loader.exec_module(ISO)


def _fits(x, y):
    return len(x) <= TMAX and len(y) <= TMAX and x.cmax() <= CMAX and y.cmax() <= CMAX and x.L() <= LMAX and y.L() <= LMAX and x.L() * y.L() <= LPROD_MAX


class Fld1:
    """Fq as Lin values"""
    lazy_out = True                                                  # results of a group operation feed the next one as they are
    def __init__(self, b):
        self.b = b
    def mul(self, a, c): return self.b.mul(a, c)
    def sqr(self, a): return self.b.mul(a, a)
    def add(self, a, c): return a + c
    def sub(self, a, c): return a - c
    def neg(self, a): return -a
    def sc(self, a, k): return a.scale(k)
    def lin(self, a, fr=False): return self.b.lin(a, fr)
    def const(self, v): return self.b.const(v) if v % Q else Lin()
    def one(self): return self.b.const(1)
    def zero(self): return Lin()
    def b3_mul(self, a, c): return self.b.mul(a.scale(4), c.scale(3))   # 3 b a c, b = 4 (g1.go: y^2 = x^3 + 4): the factor rides on the operands
    def b3_sqr(self, a): return self.b3_mul(a, a)
    def inv(self, a): return self.b.inv(a)
    def norm(self, a): return a                                      # a value that is zero iff a is
    def coords(self, a): return [a]
    def conj(self, a): return a


class Fld2:
    """Fq2 as pairs of Lin values"""
    lazy_out = False
    def __init__(self, b, T):
        self.b, self.T = b, T
    def mul(self, a, c):
        """Karatsuba (fq2.go:116-130) when the operand sums stay inside the gather bounds, schoolbook otherwise"""
        if _fits(a[0] + a[1], c[0] + c[1]):
            return self.T.mul2(a, c, kar=True)
        return self.T.mul2(a, c, kar=False)
    def sqr(self, a): return self.T.sqr2(a)
    def add(self, a, c): return Tower.add2(a, c)
    def sub(self, a, c): return Tower.sub2(a, c)
    def neg(self, a): return Tower.neg2(a)
    def sc(self, a, k): return Tower.sc2(a, k)
    def lin(self, a, fr=False): return self.T.lin2(a, fr)
    def const(self, v): return self.T.k2(v)
    def one(self): return (self.b.const(1), Lin())
    def zero(self): return (Lin(), Lin())
    def b3_mul(self, a, c):
        """3 b' a c with b' = 4 (1 + u) (g2.go): schoolbook on the operands 2a and 6c, then the factor 1 + u as a recombination"""
        return Tower.nr2(self.T.mul2(Tower.sc2(a, 2), Tower.sc2(c, 6), kar=False))
    def b3_sqr(self, a):
        """3 b' a^2 = 12 (1 + u) a^2 from two products of scaled operands: 12 (a0+a1)(a0-a1) and 24 a0 a1"""
        s, d = self.b.lin(a[0] + a[1]), self.b.lin(a[0] - a[1])
        p12, r24 = self.b.mul(s.scale(4), d.scale(3)), self.b.mul(a[0].scale(4), a[1].scale(6))
        return (p12 - r24, p12 + r24)
    def inv(self, a): return self.T.inv2(a)
    def norm(self, a):
        a = self.T.lin2(a)
        return self.b.mul(a[0], a[0]) + self.b.mul(a[1], a[1])
    def coords(self, a): return [a[0], a[1]]
    def conj(self, a): return Tower.conj2(a)


class Curve:
    def __init__(self, F):
        self.F = F

    def neg(self, P):
        return (P[0], self.F.neg(P[1]), P[2])

    def _out(self, X3, Y3, Z3):
        F = self.F
        if F.lazy_out:
            return (X3, Y3, Z3)
        return (F.lin(X3), F.lin(Y3), F.lin(Z3))

    def dbl(self, P):
        """RCB16 algorithm 9 (a = 0): with B = 3 b Z^2 and d = Y^2 - 3B,
             X3 = 2 d XY,   Y3 = 8 B Y^2 + d (Y^2 + B),   Z3 = 8 Y^3 Z                      two product levels, one recombination"""
        F = self.F
        X, Y, Z = P
        t0, t1, xy, B = F.sqr(Y), F.mul(Y, Z), F.mul(X, Y), F.b3_sqr(Z)
        d = F.sub(t0, F.sc(B, 3))
        t04 = F.sc(t0, 4)
        X3 = F.sc(F.mul(d, xy), 2)
        Y3 = F.add(F.mul(F.sc(B, 2), t04), F.mul(d, F.add(t0, B)))
        Z3 = F.mul(F.sc(t1, 2), t04)
        return self._out(X3, Y3, Z3)

    def add(self, P, R):
        """RCB16 algorithm 7 (a = 0): complete addition, 12 products in two levels"""
        F = self.F
        X1, Y1, Z1 = P
        X2, Y2, Z2 = R
        t0, t1, t2 = F.mul(X1, X2), F.mul(Y1, Y2), F.mul(Z1, Z2)
        t3 = F.lin(F.sub(F.sub(F.mul(F.add(X1, Y1), F.add(X2, Y2)), t0), t1))   # X1 Y2 + X2 Y1
        t4 = F.lin(F.sub(F.sub(F.mul(F.add(Y1, Z1), F.add(Y2, Z2)), t1), t2))   # Y1 Z2 + Y2 Z1
        B2 = F.b3_mul(Z1, Z2)                                                    # 3 b Z1 Z2
        B5 = F.lin(F.add(F.b3_mul(X1, Z2), F.b3_mul(X2, Z1)))                    # 3 b (X1 Z2 + X2 Z1)
        e = F.lin(F.sub(t1, B2)); g = F.lin(F.add(t1, B2)); t03 = F.lin(F.sc(t0, 3))
        X3 = F.sub(F.mul(t3, e), F.mul(t4, B5))
        Y3 = F.add(F.mul(e, g), F.mul(B5, t03))
        Z3 = F.add(F.mul(g, t4), F.mul(t03, t3))
        return self._out(X3, Y3, Z3)

    def mul_u64(self, P, k):
        """[k] P, most significant bit first (g1.go / g2.go Mul, same group element)"""
        R = P
        for i in range(k.bit_length() - 2, -1, -1):
            R = self.dbl(R)
            if (k >> i) & 1:
                R = self.add(R, P)
        return R

    def to_affine(self, P):
        F = self.F
        zi = F.lin(F.inv(P[2]))
        return F.lin(F.mul(P[0], zi), True), F.lin(F.mul(P[1], zi), True)


def _powers(F, x, n):
    """x^0 .. x^n with logarithmic depth"""
    pw = [F.one(), x]
    while len(pw) <= n:
        top = len(pw) - 1                                             # highest power so far: multiply the block 1..top by x^top
        base = pw[top]
        for j in range(1, top + 1):
            if len(pw) > n:
                break
            pw.append(F.lin(F.mul(base, pw[j])))
    return pw


def _poly(F, coeffs, pw):
    acc = F.const(coeffs[0])
    for k in range(1, len(coeffs)):
        c = coeffs[k]
        one = (c == 1) or (c == (1, 0))
        acc = F.add(acc, pw[k] if one else F.mul(F.const(c), pw[k]))
    return F.lin(acc)


def iso_map(F, pt, xn, xd, yn, yd):
    """the isogeny (hash.go:185-206 / 282-303) on an affine point, image in projective coordinates:
    (xn/xd, y yn/yd) = (xn yd : y yn xd : xd yd)"""
    x, y = pt
    pw = _powers(F, x, max(len(xn), len(xd), len(yn), len(yd)) - 1)
    XN, XD, YN, YD = _poly(F, xn, pw), _poly(F, xd, pw), _poly(F, yn, pw), _poly(F, yd, pw)
    return (F.lin(F.mul(XN, YD)), F.lin(F.mul(y, F.lin(F.mul(YN, XD)))), F.lin(F.mul(XD, YD)))


def _psi_consts():
    ciw = (ISO.iwsc[0], (-ISO.iwsc[1]) % Q)
    cx = f2mul(f2mul((1, 1), ciw), (ISO.kQiX, 0))
    cy = f2mul(f2mul(f2mul((1, 1), (1, 1)), ciw), (ISO.kQiY, 0))
    return cx, cy


def psi_proj(F, P):
    """psi (hash.go:341-366) on projective coordinates: (Cx conj(X) : Cy conj(Y) : conj(Z))"""
    cx, cy = _psi_consts()
    return (F.lin(F.mul(F.const(cx), F.conj(P[0]))), F.lin(F.mul(F.const(cy), F.conj(P[1]))), F.conj(P[2]))


def clear_h2_proj(C, P):
    """clearH2 (hash.go:368-389) on a projective point, same chain"""
    F = C.F
    work = C.add(C.mul_u64(P, X_ABS), P)
    mpsi = C.neg(psi_proj(F, P))
    work = C.add(work, mpsi)
    work = C.mul_u64(work, X_ABS)
    work = C.add(work, mpsi)
    work = C.add(work, C.neg(P))
    return C.add(work, psi_proj(F, psi_proj(F, C.dbl(P))))


def h2_gls_digits():
    """ScaleByCofactor (g2.go:104-115, 130-138) through clearH2 -- see gen_consts.py: [h2] P = sum (-1)^i d_i psi^i(clearH2(P))"""
    x = -X_ABS
    r = x**4 - x**2 + 1
    c = pow(3 * (x * x - 1), -1, r)
    d = [(c // X_ABS**i) % X_ABS for i in range(4)]
    assert sum(di * X_ABS**i for i, di in enumerate(d)) == c
    return d


def scale_by_cofactor_proj(C, P):
    """[h2] P = [c] Q, Q = clearH2(P), c = sum d_i |x|^i over the psi images Q_i = (-1)^i psi^i(Q); the digits are one:
    d_1 = 2 d_0 - 1, d_2 = 2 d_0 - 2, d_3 = d_0 - 1, so [h2] P = [d_0](Q_0 + 2 Q_1 + 2 Q_2 + Q_3) - (Q_1 + 2 Q_2 + Q_3) (hash.cuh: scale_by_cofactor_g2)"""
    F = C.F
    q0 = clear_h2_proj(C, P)
    p1 = psi_proj(F, q0); q2 = psi_proj(F, p1); q1 = C.neg(p1); q3 = C.neg(psi_proj(F, q2))
    d = h2_gls_digits()
    assert d[1] == 2 * d[0] - 1 and d[2] == 2 * d[0] - 2 and d[3] == d[0] - 1
    a = C.add(q1, q2)
    t = C.add(C.add(q0, q3), C.dbl(a))
    s = C.add(C.add(a, q2), q3)
    # [d_0] t by fixed 4-bit windows: the nibbles of d_0 = 0x4600 5555 5555 aaab are 0, 4, 5, 6, 10, 11 (curve.cuh: jac_mul_h2_d0)
    assert d[0] == 0x460055555555aaab
    t2 = C.dbl(t); t4 = C.dbl(t2); t5 = C.add(t4, t); t6 = C.add(t5, t); t10 = C.dbl(t5); t11 = C.add(t10, t)
    tab = {4: t4, 5: t5, 6: t6, 10: t10, 11: t11}
    res = t4
    for i in range(14, -1, -1):
        for _ in range(4):
            res = C.dbl(res)
        nib = (d[0] >> (4 * i)) & 15
        if nib:
            res = C.add(res, tab[nib])
    return C.add(res, C.neg(s))


def flat12(f):
    return [f[0][0][0], f[0][0][1], f[0][1][0], f[0][1][1], f[0][2][0], f[0][2][1],
            f[1][0][0], f[1][0][1], f[1][1][0], f[1][1][1], f[1][2][0], f[1][2][1]]


BUF_RAW3 = 8       # LOAD: element e of the raw device representation (15 int32 limbs, structure-of-arrays with n = 1) at buffer 3
BUF_RAW2 = 14      # LOAD: the same at buffer 2
BUF_M384_0 = 9     # LOAD: element e of the Fq wire format (6 little-endian uint64, Montgomery 2^384) at buffer 0
BUF_SOA3 = 10      # LOAD: coordinate e of Jacobian record w of a structure-of-arrays buffer of `stride` records (buffer 3, stride = its
                   # stride argument): element = e | w << 3 | (1 << 11 for 6-coordinate records), w < 256; a record flagged infinite reads as (0, 1, 0)


BUF_SOA12 = 11     # LOAD: Fq element e of Fq12 record (tuple index + which * arg2) of a structure-of-arrays buffer of `stride` records (buffer 3,
                   # stride = its stride argument, arg2 = buffer 2's stride argument): element = e | which << 4; a record past the end reads as 1


BUF_AFFPT = 12     # LOAD: projective coordinate e of the AFFINE wire point (tuple index + which * arg2) of buffer 0 (stride = record bytes); in_inf
                   # flags = buffer 1 (may be null); points = buffer 3's stride argument.  A flagged point, an all-zero record and a point
                   # past the end read as (0 : 1 : 0); every other as (x : y : 1).  element = e | which << 4 | (1 << 5 for G2)
BUF_SOAPT = 13     # LOAD: coordinate e of projective SoA record (tuple index + which * arg2) of buffer 3 (records = its stride argument); a record
                   # past the end reads as (0 : 1 : 0).  element = e | which << 4 | (1 << 5 for G2)


def soa_el(e, w, six):
    assert 0 <= w < 256
    return e | (w << 3) | ((1 << 11) if six else 0)


def unflat12(v):
    return (((v[0], v[1]), (v[2], v[3]), (v[4], v[5])), ((v[6], v[7]), (v[8], v[9]), (v[10], v[11])))


def build_program(kind):
    """kind: 'verify2' -- inputs P0 (buf 0, 2 Fq), Q0 (buf 1, 4 Fq), P1 (buf 2), Q1 (buf 3); verdict = FE(ML((P0,Q0),(-P1,Q1))) == 1
             'pairing1' -- inputs P (buf 0), Q (buf 1); output FE(ML(P, Q)) as 12 Fq
             'aggtail'  -- the tail of VerifyAggregate: P (buf 0), Q (buf 1) and an Fq12 R in the device representation (buf 3);
                           verdict = FE(ML(-P, Q) * R) == 1, i.e. e(P, Q) == FE(R) with ONE final exponentiation
             'aggtail2' / 'miller1rawn' -- that tail in two pieces (see below)
             'finalexp1' -- input an Fq12 in the wire format (buf 0); output FE(f) as 12 Fq (pairing.go:79-129)
             'miller1raw' -- inputs P (buf 0), Q (buf 1); output a Miller value of (P, Q) in the device representation
             'miller1x' -- inputs P (buf 0), Q (buf 1); output MillerLoop(P, Q), the reference's value (pairing.go:16-75), as 12 Fq"""
    b = Builder()
    pr = Pairing(b)
    # Every program with a final exponentiation has the squaring runs of its ExpByX ROLLED (K_REP loops, Pairing.exp_by_x): measured on one box
    # against the straight-line form (tools/rolled_ab.py, profiles/r05_rolled_ab.log): lone Pairing kernel 1.464 -> 1.459 ms, 4 096 pairings
    # 4.47 -> 4.41 ms, image 2.72 -> 1.82 MB.  'pairing1s' keeps the straight-line form of pairing1 as the A/B partner.
    pr.rolled = kind != "pairing1s"
    if kind == "pairing1s":
        kind = "pairing1"
    if kind in ("hashfin1", "hashfin2", "cofac2"):
        return build_hash_program(b, pr.T, kind)
    if kind in ("subgrp1", "subgrp2"):
        return build_subgroup_program(b, pr.T, kind)
    if kind in ("msmfin1", "msmfin2"):
        # the endomorphism MSM (msm.inc): 8 windows of 16 bits for G1 (two 128-bit halves), 4 for G2 (four 64-bit digits)
        return build_msm_final_program(b, pr.T, kind, nwin=8 if kind == "msmfin1" else 4)
    if kind in ("mul1", "mul2"):
        return build_mul_program(b, pr.T, kind)
    if kind in ("sum0_1", "sum0_2", "sum1_1", "sum1_2", "sumfin_1", "sumfin_2"):
        return build_sum_program(b, pr.T, kind)
    if kind == "aggtail":
        P = (b.inp(0, 0), -b.inp(0, 1)); Qa = ((b.inp(1, 0), b.inp(1, 1)), (b.inp(1, 2), b.inp(1, 3)))
        R = unflat12([b.inp(BUF_RAW3, e) for e in range(12)])
        R = pr.T.lin12(R, True)                                          # raw limbs of a stored value: normalised, any representative
        f = pr.final_exp(pr.T.mul12(pr.miller([(P, Qa)]), R))
        b.out = ("check1", flat12(pr.T.lin12(f, True)))
        return b
    if kind == "aggtail2":
        # the same verdict with the signature side's Miller loop taken out: S = a Miller value of (-P, Q) (program 'miller1rawn', run on a
        # side stream WHILE the tuple side is hashed and paired), R as above: verdict = FE(R * S) == 1
        R = pr.T.lin12(unflat12([b.inp(BUF_RAW3, e) for e in range(12)]), True)
        Sv = pr.T.lin12(unflat12([b.inp(BUF_RAW2, e) for e in range(12)]), True)
        f = pr.final_exp(pr.T.mul12(R, Sv))
        b.out = ("check1", flat12(pr.T.lin12(f, True)))
        return b
    if kind in ("miller1raw", "miller1rawn"):
        # MillerLoop(P, Q) for the product tree of VerifyAggregate, left in the device representation.  Its value differs from
        # the reference's Miller value by a factor in Fq2* (projective lines), which every final exponentiation downstream removes.
        # 'miller1rawn': of (-P, Q), the signature side of the aggregate's comparison
        P = (b.inp(0, 0), b.inp(0, 1) if kind == "miller1raw" else -b.inp(0, 1)); Qa = ((b.inp(1, 0), b.inp(1, 1)), (b.inp(1, 2), b.inp(1, 3)))
        b.out = ("outraw12", flat12(pr.T.lin12(pr.miller([(P, Qa)]), True)))
        return b
    if kind == "miller1x":
        # the reference's Miller value itself (MillerLoop is an exported function, pairing.go:16): reference steps, wire format
        P = (b.inp(0, 0), b.inp(0, 1)); Qa = ((b.inp(1, 0), b.inp(1, 1)), (b.inp(1, 2), b.inp(1, 3)))
        b.out = ("out12", flat12(pr.T.lin12(pr.miller([(P, Qa)], exact=True), True)))
        return b
    if kind == "powc12raw":
        # M^(1 - x), 1 - x = 1 + |x| = 0xd201000000010001: a large g2pubs VerifyAggregate pairs the hash points BEFORE their cofactor
        # clearing and raises the product of its Miller values to the cofactor multiplier once (verify_host.inc); M is an arbitrary Fq12
        # element (no final exponentiation yet), so plain squarings: 64 + 6 products.  Input and output in the device representation.
        a = pr.T.lin12(unflat12([b.inp(BUF_RAW3, e) for e in range(12)]), True)
        r = a
        c = X_ABS + 1
        for i in range(c.bit_length() - 2, -1, -1):
            r = pr.T.lin12(pr.T.sqr12(r), True)
            if (c >> i) & 1:
                r = pr.T.lin12(pr.T.mul12(r, a), True)
        b.out = ("outraw12", flat12(r))
        return b
    if kind == "mul12raw":
        # one node of the Fq12 product tree of VerifyAggregate: record t times record t + half of an SoA buffer (a missing partner
        # reads as 1), both in the device representation, result in the device representation
        a = unflat12([b.inp(BUF_SOA12, e) for e in range(12)])
        c = unflat12([b.inp(BUF_SOA12, e | 16) for e in range(12)])
        b.out = ("outraw12", flat12(pr.T.lin12(pr.T.mul12(pr.T.lin12(a, True), pr.T.lin12(c, True)), True)))
        return b
    if kind == "finalexp1":
        f = unflat12([b.inp(BUF_M384_0, e) for e in range(12)])
        b.out = ("out12", flat12(pr.T.lin12(pr.final_exp(f), True)))
        return b
    if kind == "verify1s":
        # Verify with the signature side's Miller loop taken out (small calls: it runs on a side stream while the message is hashed, program
        # 'miller1rawn'): inputs P0 (buf 0), Q0 (buf 1) and S = a Miller value of (-P1, Q1), record t of the SoA buffer 3; verdict = FE(ML(P0, Q0) * S) == 1
        P0 = (b.inp(0, 0), b.inp(0, 1)); Q0 = ((b.inp(1, 0), b.inp(1, 1)), (b.inp(1, 2), b.inp(1, 3)))
        Sv = pr.T.lin12(unflat12([b.inp(BUF_SOA12, e) for e in range(12)]), True)
        f = pr.final_exp(pr.T.mul12(pr.miller([(P0, Q0)]), Sv))
        b.out = ("check1", flat12(pr.T.lin12(f, True)))
        return b
    if kind == "verify2":
        P0 = (b.inp(0, 0), b.inp(0, 1)); Q0 = ((b.inp(1, 0), b.inp(1, 1)), (b.inp(1, 2), b.inp(1, 3)))
        P1 = (b.inp(2, 0), -b.inp(2, 1)); Q1 = ((b.inp(3, 0), b.inp(3, 1)), (b.inp(3, 2), b.inp(3, 3)))
        f = pr.final_exp(pr.miller([(P0, Q0), (P1, Q1)]))
        outs = flat12(pr.T.lin12(f, True))
        b.out = ("check1", outs)
    else:
        P = (b.inp(0, 0), b.inp(0, 1)); Qa = ((b.inp(1, 0), b.inp(1, 1)), (b.inp(1, 2), b.inp(1, 3)))
        f = pr.final_exp(pr.miller([(P, Qa)]))
        outs = flat12(pr.T.lin12(f, True))
        b.out = ("out12", outs)
    return b


def g1_beta():
    """the cube root of unity beta with phi(x, y) = (beta x, y) = [-x^2] (x, y) on G1 (see gen_consts.py: checked on the generator)"""
    G1 = (3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
          1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569)
    def add(P1, P2):
        if P1 is None: return P2
        if P2 is None: return P1
        (x1, y1), (x2, y2) = P1, P2
        if x1 == x2:
            if (y1 + y2) % Q == 0: return None
            l = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
        else:
            l = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
        x3 = (l * l - x1 - x2) % Q
        return (x3, (l * (x1 - x3) - y1) % Q)
    r = X_ABS**4 - X_ABS**2 + 1
    lam = (-(X_ABS ** 2)) % r
    R = None
    for bit in bin(lam)[2:]:
        R = add(R, R)
        if bit == "1": R = add(R, G1)
    for g in range(2, 20):
        beta = pow(g, (Q - 1) // 3, Q)
        if beta != 1 and (beta * G1[0] % Q, G1[1]) == R:
            return beta
    raise AssertionError("beta")


def build_subgroup_program(b, T, kind):
    """IsInCorrectSubgroupAssumingOnCurve (g1.go:137-141, g2.go:293-295) through the endomorphisms, as k_hash.hip does it per
    lane: G1: [x^2] P + phi(P) is the point at infinity; G2: [|x|] P + psi(P) is.  Output: the Z coordinate of that sum (zero
    exactly for points of the subgroup, the formulas being complete).  Input: an affine point ON the curve (buffer 0)."""
    if kind == "subgrp1":
        F = Fld1(b); C = Curve(F)
        x, y = b.inp(0, 0), b.inp(0, 1)
        one = F.one()
        P = (x, y, one)
        R = C.mul_u64(C.mul_u64(P, X_ABS), X_ABS)
        S = C.add(R, (F.lin(F.mul(F.const(g1_beta()), x)), y, one))
        outs = [b.lin(S[2], True)]
    else:
        F = Fld2(b, T); C = Curve(F)
        pt = ((b.inp(0, 0), b.inp(0, 1)), (b.inp(0, 2), b.inp(0, 3)))
        P = (pt[0], pt[1], F.one())
        S = C.add(C.mul_u64(P, X_ABS), psi_proj(F, P))
        z = F.lin(S[2], True)
        outs = [z[0], z[1]]
    b.out = ("iszero", outs, len(outs))
    return b


def build_msm_final_program(b, T, kind, nwin, c=16, m=13, logk=3):
    """the tail of the bucket-method MSM over decomposed scalars (msm.inc): per window w the fold leaves 2 + m sums -- X (all chunk
    sums), L (the chunks' local weighted sums) and the odd-element sums O_0 .. O_{m-1} of the successively halved chunk-sum arrays --
    at records a nwin + w of a Jacobian structure-of-arrays buffer (a = 0: X, 1: L, 2 + l: O_l).  The window total is
        L - X + 2^logk (O_0 + 2 O_1 + 4 O_2 + ...)                     (Horner from O_{m-1} down, all windows side by side)
    and the windows are joined by Horner in 2^c -- c doublings and one addition per window -- then converted to affine.
    (X, Y, Z) Jacobian = (X Z : Y : Z^3) homogeneous.  Output: the affine sum and its Z (zero for the point at infinity)."""
    six = kind == "msmfin2"
    F = Fld2(b, T) if six else Fld1(b)
    C = Curve(F)
    # every input is loaded -- and converted -- BEFORE the first chain is written down: the scheduler places a job at the earliest
    # level of its kind with room and opens new levels only at the end, so a load that no longer fits an early LOAD level would land
    # behind everything created so far and serialise the windows
    nrec = (m + 2) * nwin
    if six:
        raw = [[(b.inp(BUF_SOA3, soa_el(2 * j, rec, True)), b.inp(BUF_SOA3, soa_el(2 * j + 1, rec, True))) for j in range(3)] for rec in range(nrec)]
        red = [[(b.lin(c[0], True), b.lin(c[1], True)) for c in r] for r in raw]
    else:
        raw = [[b.inp(BUF_SOA3, soa_el(j, rec, False)) for j in range(3)] for rec in range(nrec)]
        red = [[b.lin(c, True) for c in r] for r in raw]
    z2s = [F.lin(F.sqr(r[2])) for r in red]
    pts = [(F.lin(F.mul(r[0], r[2])), r[1], F.lin(F.mul(z2, r[2]))) for r, z2 in zip(red, z2s)]
    def point(rec):
        return pts[rec]
    W = [point((m + 1) * nwin + w) for w in range(nwin)]
    for l in range(m - 2, -1, -1):                                          # all windows step by step: their chains share levels
        W = [C.add(C.dbl(W[w]), point((l + 2) * nwin + w)) for w in range(nwin)]
    for _ in range(logk):
        W = [C.dbl(t) for t in W]
    W = [C.add(W[w], point(nwin + w)) for w in range(nwin)]
    W = [C.add(W[w], C.neg(point(w))) for w in range(nwin)]
    R = W[nwin - 1]
    for w in range(nwin - 2, -1, -1):
        for _ in range(c):
            R = C.dbl(R)
        R = C.add(R, W[w])
    x, y = C.to_affine(R)
    outs = F.coords(x) + F.coords(y)
    b.out = ("outaff", outs + [b.lin(F.norm(R[2]), True)], len(outs))
    return b


def build_mul_program(b, T, kind):
    """[k] P for a run-time 256-bit scalar k and P in the prime-order subgroup (g1.go:80-90 / g2.go MulFR: same group element),
    through the curve endomorphisms (glv_model.py): the scalar arrives DECOMPOSED, as the 64-byte big-endian digit record a small
    kernel (k_glv_recode) writes -- G1: k1 in bits [0, 256), k2 in [256, 512) with k = k1 + k2 z^2; G2: the base-z digits d_i in
    bits [128 i, 128 i + 128) -- and every sub-scalar walks fixed 4-bit windows over its own table: 0 P .. 15 P and the images
    of those entries under -phi (G1) / -psi, psi^2, -psi^3 (G2).  A SEL level picks an entry per table and window; the picked
    entries are summed (one / two addition levels, beside the four doublings of the accumulator) and join the accumulator
    with one addition: 33 (G1) / 17 (G2) windows instead of 64, the same two product levels per doubling.  The control flow
    does not depend on the scalar.  Inputs: the affine point (buffer 0), the digit record (buffer 1).
    Output: the affine product and its Z (zero for the point at infinity)."""
    import glv_model as GLV
    six = kind == "mul2"
    F = Fld2(b, T) if six else Fld1(b)
    C = Curve(F)
    if six:
        pt = ((b.inp(0, 0), b.inp(0, 1)), (b.inp(0, 2), b.inp(0, 3)))
    else:
        pt = (b.inp(0, 0), b.inp(0, 1))
    P = (pt[0], pt[1], F.one())
    tab = [(F.zero(), F.one(), F.zero()), P]
    for d in range(2, 16):
        tab.append(C.dbl(tab[d // 2]) if d % 2 == 0 else C.add(tab[d - 1], P))
    tab = [tuple(F.lin(c) for c in q) for q in tab]
    flat = (lambda q: [q[0][0], q[0][1], q[1][0], q[1][1], q[2][0], q[2][1]]) if six else (lambda q: [q[0], q[1], q[2]])
    unflat = (lambda v: ((v[0], v[1]), (v[2], v[3]), (v[4], v[5]))) if six else (lambda v: (v[0], v[1], v[2]))
    if six:
        t1 = [C.neg(psi_proj(F, q)) for q in tab]                                       # -psi
        t2 = [psi_proj(F, psi_proj(F, q)) for q in tab]                                 # psi^2
        t3 = [C.neg(psi_proj(F, q)) for q in t2]                                        # -psi^3
        tables, nwin, stride = [tab, t1, t2, t3], GLV.G2_LAT_NWIN, 32                   # digit i: bits [128 i, 128 i + 68) = windows 32 i ..
    else:
        beta = F.const(g1_beta())
        t1 = [(F.lin(F.mul(beta, q[0])), F.neg(q[1]), q[2]) for q in tab]               # -phi (X : Y : Z) = (beta X : -Y : Z)
        tables, nwin, stride = [tab, t1], GLV.G1_LAT_NWIN, 64
    tids = [b.table([flat(q) for q in t]) for t in tables]
    R = None
    for w in range(nwin - 1, -1, -1):
        if R is not None:
            for _ in range(4):
                R = C.dbl(R)
        S = [unflat(b.select(tid, i * stride + w)) for i, tid in enumerate(tids)]
        U = C.add(S[0], S[1])
        if six:
            U = C.add(U, C.add(S[2], S[3]))
        R = U if R is None else C.add(R, U)
    x, y = C.to_affine(R)
    outs = F.coords(x) + F.coords(y)
    b.out = ("outaff", outs + [b.lin(F.norm(R[2]), True)], len(outs))
    return b


def build_sum_program(b, T, kind):
    """the tree sum of affine points (AggregateSignatures / AggregatePublicKeys g2pubs/bls.go:165-192, the tail of small MSMs) for
    small counts, one addition per wave with the complete projective formulas (the point at infinity needs no flag):
      'sum0_g' -- level 0: wire points t and t + half (flags / zero records / a missing partner = infinity) -> projective sum
      'sum1_g' -- inner levels: projective records t and t + half -> their sum
      'sumfin_g' -- the root: one projective record -> affine wire bytes and its Z (zero for the point at infinity)"""
    six = kind.endswith("2")
    F = Fld2(b, T) if six else Fld1(b)
    C = Curve(F)
    bit = 32 if six else 0
    def point(buf, which):
        def el(e):
            return b.lin(b.inp(buf, e | (which << 4) | bit), True) if buf == BUF_SOAPT else b.inp(buf, e | (which << 4) | bit)
        if six:
            return ((el(0), el(1)), (el(2), el(3)), (el(4), el(5)))
        return (el(0), el(1), el(2))
    if kind.startswith("sumfin"):
        R = point(BUF_SOAPT, 0)
        x, y = C.to_affine(R)
        outs = F.coords(x) + F.coords(y)
        b.out = ("outaff", outs + [b.lin(F.norm(R[2]), True)], len(outs))
        return b
    buf = BUF_AFFPT if kind.startswith("sum0") else BUF_SOAPT
    R = C.add(point(buf, 0), point(buf, 1))
    outs = []
    for c in R:
        outs += [b.lin(v, True) for v in F.coords(c)]
    b.out = ("outraw12", outs)
    return b


def build_hash_program(b, T, kind):
    """the curve-arithmetic tails of hash-to-curve; inputs: the affine outputs of the SWU maps (buffer 0), output: the affine
    hash point as wire-format field elements plus values that are zero exactly when a step was exceptional (an isogeny
    denominator vanished, the two mapped points have the same x, the result is the point at infinity) -- the caller then
    takes the one-message-per-lane kernels, which follow the reference's steps literally.
      'hashfin1' -- HashG1 after the two SWU maps (hash.go:311-321): iso11 of both points, sum, clearH = [|x| + 1]
      'hashfin2' -- HashG2 after the two SWU maps (hash.go:391-402): iso3 of both points, sum, clearH2
      'cofac2'   -- HashG2WithDomain after the try-and-increment search (g2.go:1078-1084): ScaleByCofactor of the affine point
    The reference adds the two mapped points first and applies the isogeny to the sum; the isogeny is a group homomorphism,
    so mapping each point and adding the images is the same point -- and lets both maps share their levels."""
    if kind == "hashfin1":
        F = Fld1(b); C = Curve(F)
        pts = [(b.inp(0, 0), b.inp(0, 1)), (b.inp(0, 2), b.inp(0, 3))]
        im = [iso_map(F, q, ISO.xNum11, ISO.xDen11, ISO.yNum11, ISO.yDen11) for q in pts]
        s = C.add(im[0], im[1])
        r = C.add(C.mul_u64(s, X_ABS), s)
    elif kind == "hashfin2":
        F = Fld2(b, T); C = Curve(F)
        pts = [((b.inp(0, 0), b.inp(0, 1)), (b.inp(0, 2), b.inp(0, 3))), ((b.inp(0, 4), b.inp(0, 5)), (b.inp(0, 6), b.inp(0, 7)))]
        im = [iso_map(F, q, ISO.xNum3, ISO.xDen3, ISO.yNum3, ISO.yDen3) for q in pts]
        r = clear_h2_proj(C, C.add(im[0], im[1]))
    else:
        F = Fld2(b, T); C = Curve(F)
        pt = ((b.inp(0, 0), b.inp(0, 1)), (b.inp(0, 2), b.inp(0, 3)))
        im = []
        r = scale_by_cofactor_proj(C, (pt[0], pt[1], F.one()))
    x, y = C.to_affine(r)
    checks = [b.lin(F.norm(r[2]), True)] + [b.lin(F.norm(q[2]), True) for q in im]
    if im:
        # equal or opposite mapped points: the reference's addition then doubles with the a = 0 formula on a curve whose a is
        # not 0 (g1.go / g2.go AddMixed -> Double), or stops at infinity; neither is reproduced here, both are flagged
        checks.append(b.lin(F.norm(F.sub(pts[0][0], pts[1][0])), True))
    outs = F.coords(x) + F.coords(y)
    b.out = ("outaff", outs + checks, len(outs))
    return b


# ---- levelling, slot allocation ---------------------------------------------------------------------------------------
class Program:
    pass


def schedule(b):
    """as-soon-as-possible levels of one kind each (<= 64 jobs, INV: 1), then slots by live range.  Dead nodes are dropped."""
    outs = b.out[1]
    for o in outs:
        assert len(o) == 1 and list(o.values()) == [1], "outputs must be materialised"
    out_nodes = [list(o.keys())[0] for o in outs]
    # liveness: mark reachable
    live = set()
    stack = list(out_nodes)
    while stack:
        n = stack.pop()
        if n.id in live:
            continue
        live.add(n.id)
        for lin in (n.x, n.y):
            if lin:
                stack.extend(lin.keys())
    nodes = [n for n in b.nodes if n.id in live]
    levels = []                          # [kind, [nodes]]
    kind_of = {"mul": K_MUL, "lin": K_LIN, "inv": K_INV, "in": K_LOAD, "sel": K_SEL, "sqr3": K_SQR}
    cap = {K_MUL: LANES, K_LIN: LANES, K_INV: 1, K_LOAD: LANES, K_SEL: LANES, K_SQR: LANES}
    consts = [n for n in nodes if n.kind == "const"]
    for n in consts:
        n.level = -1
    ring_level, exclusive = {}, {}
    for n in nodes:
        if n.kind == "const":
            continue
        k = kind_of[n.kind]
        e = 0
        for lin in (n.x, n.y):
            if lin:
                for d in lin:
                    e = max(e, d.level + 1)
        lv = None
        if n.ring is not None:                                             # a rolled iteration's jobs: a level of their own (nothing else may join it: the level must repeat byte for byte)
            tag = (n.ring[0], n.ring[1], k)
            lv = ring_level.get(tag)
            if lv is None:
                levels.append([k, []])
                lv = len(levels) - 1
                ring_level[tag] = lv; exclusive[lv] = tag
            assert lv >= e
        else:
            for i in range(e, len(levels)):
                if i not in exclusive and levels[i][0] == k and len(levels[i][1]) < cap[k]:
                    lv = i
                    break
        if lv is None:
            levels.append([k, []])
            lv = len(levels) - 1
            assert lv >= e
        levels[lv][1].append(n)
        n.level = lv
    nlev = len(levels)
    # last use
    for n in nodes:
        for lin in (n.x, n.y):
            if lin:
                for d in lin:
                    d.last = max(d.last, n.level)
    for n in out_nodes:
        n.last = nlev                    # read by the output level
    # slots: constants first (live for ever), then linear scan; a slot freed at level l (last read at l) is reusable by a
    # job WRITING at level >= l (reads of a level precede its writes in the wave's program order)
    nslot = 0
    for n in consts:
        n.slot = nslot; nslot += 1
    table_base = {}
    for tid, (ncoord, tnodes) in enumerate(getattr(b, "tables", [])):     # tables: contiguous blocks, never recycled
        table_base[tid] = nslot
        for n in tnodes:
            n.slot = nslot + n.pin[1]
        nslot += len(tnodes)
    # slots by live range.  A rolled run's 54 fixed slots are taken from the FREE list at the run's first level and go back to it one by one, each when
    # its last value is dead: a run borrows what the products around it have just released, and costs (almost) no LDS of its own.
    ring_nodes = [n for n in nodes if n.ring is not None]
    run_first = {}
    for n in ring_nodes:
        run_first[n.ring[0]] = min(run_first.get(n.ring[0], n.level), n.level)
    starts, ends = {}, {}
    for rid, lv in run_first.items():
        starts.setdefault(lv, []).append(rid)
    slot_last = {}
    for n in ring_nodes:                                                   # every fixed slot goes back as soon as ITS last value is dead
        key = (n.ring[0], n.ring[2])
        slot_last[key] = max(slot_last.get(key, -1), n.last)
    for key, lv in slot_last.items():
        ends.setdefault(lv, []).append(key)
    ring_slots = {}
    free = []
    expiring = {}
    for n in nodes:
        if n.kind != "const" and not n.pin and n.ring is None:
            expiring.setdefault(n.last, []).append(n)
    for li, (k, jobs) in enumerate(levels):
        for n in expiring.get(li - 1, []):  # values last read in an EARLIER level are dead (a multi-wave workgroup gathers and
            free.append(n.slot)             # stores of one level without a barrier in between: no reuse within the level)
        for rid, idx in ends.get(li - 1, []):
            free.append(ring_slots[rid][idx])
        for rid in starts.get(li, []):
            got = []
            while len(got) < 54:
                if free:
                    got.append(free.pop())
                else:
                    got.append(nslot); nslot += 1
            ring_slots[rid] = got
        for n in jobs:
            if n.pin:
                continue
            if n.ring is not None:
                n.slot = ring_slots[n.ring[0]][n.ring[2]]
                continue
            if free:
                n.slot = free.pop()
            else:
                n.slot = nslot; nslot += 1
    by_slot = {}
    for n in ring_nodes:
        by_slot.setdefault(n.slot, []).append(n)
    for lst in by_slot.values():                                           # users of one fixed slot must not overlap: written strictly after the previous value's last read
        lst.sort(key=lambda n: n.level)
        for a, c in zip(lst, lst[1:]):
            assert c.level > a.last, ("rolled run: slot reused while live", a.ring, a.level, a.last, c.ring, c.level)
    p = Program()
    p.levels, p.consts, p.nslot, p.out, p.out_nodes, p.nodes = levels, consts, nslot, b.out[0], out_nodes, nodes
    # loops: per rolled run, iterations 1 .. n - 2 sit in consecutive exclusive levels (SQR, LIN each); (1, 2), (3, 4), ... repeat
    p.repeats = []
    runs = {}
    for lv, (rid, it, k) in exclusive.items():
        runs.setdefault(rid, {}).setdefault(it, []).append(lv)
    for rid, its in sorted(runs.items()):
        nit = max(its) + 1
        count = (nit - 1) // 2                                             # iterations 1 .. nit - 1 in pairs
        if count >= 2:
            start = min(its[1])
            for j in range(2 * count):
                assert sorted(its[1 + j]) == [start + 2 * j, start + 2 * j + 1], "rolled run: iteration levels are not consecutive"
            p.repeats.append((start, 4, count))
    p.nout = b.out[2] if len(b.out) > 2 else len(out_nodes)
    p.table_base = table_base
    p.table_ncoord = {tid: nc for tid, (nc, _) in enumerate(getattr(b, "tables", []))}
    return p


# ---- exact simulator (field arithmetic on Python integers, slots reused exactly as scheduled) --------------------------
def simulate(p, inputs):
    """inputs: {buf: [Fq ints]} -> list of 12 output values (normal form)"""
    S = [None] * p.nslot
    for n in p.consts:
        S[n.slot] = n.aux
    def ev(lin):
        return sum(c * S[d.slot] for d, c in lin.items()) % Q
    for k, jobs in p.levels:
        res = []
        for n in jobs:
            if n.kind == "in":
                res.append(inputs[n.aux[0]][n.aux[1]] % Q)
            elif n.kind == "mul":
                res.append(ev(n.x) * ev(n.y) % Q)
            elif n.kind == "lin":
                res.append(ev(n.x))
            elif n.kind == "sqr3":
                v = ev(n.x); res.append(3 * v * v % Q)
            elif n.kind == "inv":
                v = ev(n.x)
                res.append(pow(v, -1, Q) if v else 0)
            elif n.kind == "sel":
                tid, j, w = n.aux
                digit = (inputs["scalar"] >> (4 * w)) & 15
                res.append(S[p.table_base[tid] + digit * p.table_ncoord[tid] + j])
        for n, r in zip(jobs, res):
            S[n.slot] = r
    return [S[n.slot] for n in p.out_nodes]


# ---- binary image ---------------------------------------------------------------------------------------------------------
def mont_limbs(v):
    v = v * RMONT % Q
    return [(v >> (LB * i)) & ((1 << LB) - 1) for i in range(NLIMB)]


def encode(p):
    """header: magic, nlevels, nslot, nconst, out kind; per level: kind, ntx, nty, njobs (LIN: lanes per job); per (level, lane): 16 x u16
    [dst, 7 x-terms, 7 y-terms, flags]; term = slot | (coef + 16) << 11; constants: slot + 15 limbs."""
    assert p.nslot < 2048
    def term(d, c):
        assert -CMAX <= c <= CMAX and c != 0
        return d.slot | ((c + 16) << 11)
    hdr = []
    desc = bytearray()
    NOTERM = 16 << 11                       # coefficient 0, slot 0
    dummy = p.nslot                         # idle lanes write to a spare slot
    rep_at = {start: (ln, cnt) for start, ln, cnt in getattr(p, "repeats", [])}
    body = None                             # (first level, length, count, encodings of the first pass) while inside a rolled run
    for li, (k, jobs) in enumerate(p.levels):
        if li in rep_at:
            ln, cnt = rep_at[li]
            hdr.append((K_REP, ln, cnt, 0))
            body = [li, ln, cnt, []]
        ntx = nty = 0
        rows = []
        # A LIN level rarely has more than 16 jobs: its jobs are then spread over 4 (or 2) adjacent lanes each, every lane gathers
        # a quarter of the terms and the partial sums are added across the lanes (DPP) before the normalisation.
        split = (4 if len(jobs) <= 16 else 2 if len(jobs) <= 32 else 1) if k == K_LIN else 1
        for n in jobs:
            xs = [term(d, c) for d, c in n.x.items()] if n.x else []
            ys = [term(d, c) for d, c in n.y.items()] if n.y else []
            flags = 0
            if k == K_LIN and split > 1:
                assert len(xs) <= TLIN
                flags = 1 if n.reduce else 0
                for q in range(split):
                    part = xs[q::split]
                    ntx, nty = max(ntx, len(part[:7])), max(nty, len(part[7:]))
                    rows.append([n.slot if q == 0 else dummy] + part[:7] + [NOTERM] * (7 - len(part[:7])) + part[7:] + [NOTERM] * (7 - len(part[7:])) + [flags])
                continue
            if k == K_LIN:
                assert len(xs) <= TLIN
                xs, ys = xs[:7], xs[7:]
                flags = 1 if n.reduce else 0
            elif k == K_LOAD:
                xs = [n.aux[0] | (n.aux[1] << 4)]; ys = []
            elif k == K_SEL:                                             # raw fields: slot of entry 0, window; stride
                tid, j, w = n.aux
                xs = [p.table_base[tid] + j, w]; ys = [p.table_ncoord[tid], 63]   # 63: the last byte of the 64-byte digit record holds windows 0, 1
            assert len(xs) <= 7 and len(ys) <= 7
            ntx, nty = max(ntx, len(xs)), max(nty, len(ys))
            rows.append([n.slot] + xs + [NOTERM] * (7 - len(xs)) + ys + [NOTERM] * (7 - len(ys)) + [flags])
        while len(rows) < LANES:
            rows.append([dummy] + [NOTERM] * 14 + [0])
        red = 0x80 if (k == K_LIN and any(n.reduce for n in jobs)) else 0   # value reduction is level-wide (always valid, never needed less)
        h = (k | red, ntx, nty, split if k == K_LIN else len(jobs))           # last byte: LIN: lanes per job; otherwise the job count
        enc = b"".join(struct.pack("<16H", *r) for r in rows)
        if body is not None:
            first, ln, cnt, seen = body
            j = li - first
            if j < ln:
                seen.append((h, enc))                                      # first pass: emitted
            else:
                assert (h, enc) == seen[j % ln], "rolled run: level %d differs from its first pass" % li   # later passes: must BE the first pass
                if j == ln * cnt - 1:
                    body = None
                continue
            if ln * cnt == ln:
                body = None
        desc += enc
        hdr.append(h)
    out_kind = {"check1": K_CHECK1, "out12": K_OUT12, "outraw12": K_OUTRAW12, "outaff": K_OUTAFF, "iszero": K_ISZERO}[p.out]
    blob = bytearray()
    assert len(p.out_nodes) <= 12
    blob += struct.pack("<8I", 0x54414c42, len(hdr), p.nslot + 1, len(p.consts), out_kind, p.nout, len(p.out_nodes) - p.nout, 0)   # (levels as stored: a rolled run counts once, plus its K_REP marker)
    blob += struct.pack("<12H", *([n.slot for n in p.out_nodes] + [0] * (12 - len(p.out_nodes)))) + b"\0" * 8
    for h in hdr:
        blob += struct.pack("<4B", *h)
    while len(blob) % 16:
        blob += b"\0"
    for n in p.consts:
        blob += struct.pack("<16i", *(mont_limbs(n.aux) + [n.slot]))
    blob += desc
    return bytes(blob)


def executed_levels(blob):
    """the level stream as the kernel EXECUTES it: [(header 4-tuple, the level's 2 048 descriptor bytes)], K_REP loops expanded -- the decoder's view of
    encode()'s output (tests/test_lat_program.py compares it with the straight-line encoding of the same schedule)"""
    magic, nhdr, nslot, nconst = struct.unpack_from("<4I", blob, 0)
    assert magic == 0x54414c42
    off = 32 + 24 + 8
    hdr = [struct.unpack_from("<4B", blob, off + 4 * i) for i in range(nhdr)]
    pos = off + 4 * nhdr
    pos += (-pos) % 16
    pos += 64 * nconst
    out, l, dl = [], 0, 0
    rep_left, rep_lo, rep_hi, rep_dlo = 0, 0, None, 0
    while l < nhdr:
        h = hdr[l]
        if h[0] & 0x7f == K_REP:
            rep_left, rep_lo, rep_hi, rep_dlo = h[2], l + 1, l + 1 + h[1], dl
            l += 1
            continue
        out.append((h, blob[pos + 2048 * dl: pos + 2048 * (dl + 1)]))
        l += 1; dl += 1
        if rep_hi is not None and l == rep_hi:
            if rep_left > 1:
                rep_left -= 1; l, dl = rep_lo, rep_dlo
            else:
                rep_hi = None
    assert pos + 2048 * dl == len(blob), "descriptor rows left over"
    return out


def stats(p):
    from collections import Counter
    c = Counter(k for k, _ in p.levels)
    jobs = Counter()
    for k, j in p.levels:
        jobs[k] += len(j)
    return "levels=%d (mul %d, sqr %d, lin %d, inv %d, load %d)  jobs: mul %d lin %d  slots=%d consts=%d" % (
        len(p.levels), c[K_MUL], c[K_SQR], c[K_LIN], c[K_INV], c[K_LOAD], jobs[K_MUL], jobs[K_LIN], p.nslot, len(p.consts))


def main():
    import pathlib
    here = pathlib.Path(__file__).resolve().parent
    sys.setrecursionlimit(100000)
    out = bytearray()
    index = []
    for name in ("verify2", "verify1s", "pairing1", "pairing1s", "aggtail", "aggtail2", "finalexp1", "miller1raw", "miller1rawn", "miller1x", "hashfin1", "hashfin2", "cofac2", "subgrp1", "subgrp2", "msmfin1", "msmfin2", "mul1", "mul2", "mul12raw", "powc12raw", "sum0_1", "sum0_2", "sum1_1", "sum1_2", "sumfin_1", "sumfin_2"):
        p = schedule(build_program(name))
        blob = encode(p)
        index.append((name, len(out), len(blob)))
        out += blob
        while len(out) % 256:
            out += b"\0"
        print(name, stats(p), "bytes=%d" % len(blob))
    (here / "lat_programs.bin").write_bytes(bytes(out))
    import zlib
    (here / "lat_programs.z").write_bytes(zlib.compress(bytes(out), 9))    # what libblsmi.so embeds (inflated once at start-up): 19 MB -> 1.3 MB
    with open(here / "lat_programs.h", "w") as f:
        f.write("// Generated by gen_lat.py -- offsets of the latency-path programs inside lat_programs.bin\n#pragma once\n")
        for name, off, ln in index:
            f.write("#define LAT_%s_OFFSET %d\n#define LAT_%s_BYTES %d\n" % (name.upper(), off, name.upper(), ln))
        f.write("#define LAT_TOTAL_BYTES %d\n" % len(out))


if __name__ == "__main__":
    main()
