"""CPU (no GPU): the level programs of the latency path (bls_amd/csrc/gen_lat.py) evaluated by the generator's exact
big-integer simulator -- slot reuse as scheduled -- against the oracle's pairing: FinalExponentiation(MillerLoop) bit for
bit for one pair, and the CompareTwoPairings verdict/value for two pairs (pairing.go:132-147).  Also the bound accounting
the kernel's integer arithmetic relies on (fp.cuh: limb bound L, value bound V per gathered operand)."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_lat", os.path.join(ROOT, "bls_amd", "csrc", "gen_lat.py"))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)
from oracle import pyref as P  # noqa: E402

sys.setrecursionlimit(100000)


@pytest.fixture(scope="module")
def progs():
    return {k: G.schedule(G.build_program(k)) for k in ("pairing1", "verify2", "verify1s", "aggtail", "aggtail2", "powc12raw", "miller1rawn", "finalexp1", "miller1x", "hashfin1", "hashfin2", "cofac2", "subgrp1", "subgrp2")}


def _pt(xs):
    a, b = P.rand_fr(xs), P.rand_fr(xs)
    return P.jac_to_affine(P.F1, P.affine_mul(P.F1, P.G1_GEN, a)), P.jac_to_affine(P.F2, P.affine_mul(P.F2, P.G2_GEN, b))


def test_pairing_program_matches_oracle(progs, kats):
    p = progs["pairing1"]
    xs = P.XORShift(5)
    for Pa, Qa in [(P.G1_GEN, P.G2_GEN), _pt(xs), _pt(xs)]:
        out = G.simulate(p, {0: [Pa[0], Pa[1]], 1: [Qa[0][0], Qa[0][1], Qa[1][0], Qa[1][1]]})
        assert out == P.fq12_flat(P.pairing(Pa, Qa))
    out = G.simulate(p, {0: list(P.G1_GEN), 1: [P.G2_GEN[0][0], P.G2_GEN[0][1], P.G2_GEN[1][0], P.G2_GEN[1][1]]})
    assert out == [int(v) for v in kats["pairing_g1gen_g2gen"]]               # the reference's own vector, pairing_test.go:9-58


def test_rolled_pairing_program(kats):
    """pairing1 (and every program with a final exponentiation) has the squaring runs of its five ExpByX (pairing.go:92-98: the reference writes them as loops, fq12.go:112-118)
    ROLLED: fixed ping-pong slots, a value reduction on every even iteration, the kernel repeats four levels (K_REP).  (i) the schedule with
    its fixed slots computes the reference's pairing (exact simulator); (ii) what the kernel executes -- the encoding with its loops expanded --
    is byte for byte the straight-line encoding of the same schedule; (iii) the image is a third smaller."""
    p = G.schedule(G.build_program("pairing1"))
    assert len(p.repeats) == 15 and sum(c for _, _, c in p.repeats) == 5 * (3 + 15 + 7) - 1      # runs of 9, 32, 16 squarings in |x|; 8, 32, 15 in |x| >> 1
    xs = P.XORShift(6)
    for Pa, Qa in [(P.G1_GEN, P.G2_GEN), _pt(xs)]:
        out = G.simulate(p, {0: [Pa[0], Pa[1]], 1: [Qa[0][0], Qa[0][1], Qa[1][0], Qa[1][1]]})
        assert out == P.fq12_flat(P.pairing(Pa, Qa))
    out = G.simulate(p, {0: list(P.G1_GEN), 1: [P.G2_GEN[0][0], P.G2_GEN[0][1], P.G2_GEN[1][0], P.G2_GEN[1][1]]})
    assert out == [int(v) for v in kats["pairing_g1gen_g2gen"]]
    rolled = G.encode(p)
    reps, p.repeats = p.repeats, []
    straight = G.encode(p)
    p.repeats = reps
    a, b = G.executed_levels(rolled), G.executed_levels(straight)
    assert len(a) == len(b) == len(p.levels) and a == b
    assert len(rolled) < 0.7 * len(straight)
    # the straight-line program next to it: the same number of executed levels (rolling costs no level), fewer slots
    q = G.schedule(G.build_program("pairing1s"))
    assert not q.repeats and len(q.levels) == len(p.levels) and q.nslot + 54 >= p.nslot - 2


def test_exact_miller_program_is_the_reference_miller_value(progs):
    p = progs["miller1x"]
    xs = P.XORShift(15)
    for Pa, Qa in [(P.G1_GEN, P.G2_GEN), _pt(xs)]:
        out = G.simulate(p, {0: [Pa[0], Pa[1]], 1: [Qa[0][0], Qa[0][1], Qa[1][0], Qa[1][1]]})
        assert out == P.fq12_flat(P.miller_loop([(Pa, P.g2_prepare(Qa))]))              # pairing.go:16-75


def test_verify_program_matches_compare_two_pairings(progs):
    p = progs["verify2"]
    xs = P.XORShift(6)
    a = P.rand_fr(xs)
    P0 = P.jac_to_affine(P.F1, P.affine_mul(P.F1, P.G1_GEN, a)); Q0 = P.G2_GEN
    P1 = P.G1_GEN; Q1 = P.jac_to_affine(P.F2, P.affine_mul(P.F2, P.G2_GEN, a))

    def run(P0, Q0, P1, Q1):
        return G.simulate(p, {0: [P0[0], P0[1]], 1: [Q0[0][0], Q0[0][1], Q0[1][0], Q0[1][1]], 2: [P1[0], P1[1]], 3: [Q1[0][0], Q1[0][1], Q1[1][0], Q1[1][1]]})
    assert run(P0, Q0, P1, Q1) == [1] + [0] * 11 and P.compare_two_pairings(P0, Q0, P1, Q1)
    Q1b = P.jac_to_affine(P.F2, P.affine_mul(P.F2, P.G2_GEN, a + 1))
    f = P.final_exponentiation(P.miller_loop([(P0, P.g2_prepare(Q0)), (P.affine_neg(P.F1, P1), P.g2_prepare(Q1b))]))
    assert run(P0, Q0, P1, Q1b) == P.fq12_flat(f) != [1] + [0] * 11
    # the same verdicts with the second pair's Miller loop run apart (programs miller1rawn + verify1s: small calls, side stream)
    for q1, want in ((Q1, True), (Q1b, False)):
        Sv = G.simulate(progs["miller1rawn"], {0: [P1[0], P1[1]], 1: [q1[0][0], q1[0][1], q1[1][0], q1[1][1]]})
        out = G.simulate(progs["verify1s"], {0: [P0[0], P0[1]], 1: [Q0[0][0], Q0[0][1], Q0[1][0], Q0[1][1]], G.BUF_SOA12: Sv})
        assert (out == [1] + [0] * 11) is want


def test_final_exponentiation_and_aggregate_tail_programs(progs):
    xs = P.XORShift(9)
    a = P.rand_fr(xs)
    Pa = P.jac_to_affine(P.F1, P.affine_mul(P.F1, P.G1_GEN, a)); Qa = P.G2_GEN
    ml = P.miller_loop([(Pa, P.g2_prepare(Qa))])
    assert G.simulate(progs["finalexp1"], {G.BUF_M384_0: P.fq12_flat(ml)}) == P.fq12_flat(P.final_exponentiation(ml))     # pairing.go:79-129
    # aggregate tail: e(P, Q) == FE(R) decided as FE(ML(-P, Q) * R) == 1, with R any Miller value of an equal / unequal pairing
    qin = {0: [Pa[0], Pa[1]], 1: [Qa[0][0], Qa[0][1], Qa[1][0], Qa[1][1]]}
    for da, want in ((0, True), (1, False)):
        Q1 = P.jac_to_affine(P.F2, P.affine_mul(P.F2, P.G2_GEN, a + da))
        R = P.miller_loop([(P.G1_GEN, P.g2_prepare(Q1))])
        inputs = dict(qin); inputs[G.BUF_RAW3] = P.fq12_flat(R)
        out = G.simulate(progs["aggtail"], inputs)
        assert (out == [1] + [0] * 11) is want
        # the tail in two pieces (one-context calls): S = a Miller value of (-P, Q) from the side stream, then FE(R * S) == 1
        Sv = G.simulate(progs["miller1rawn"], qin)
        assert P.fq12_flat(P.final_exponentiation(G.unflat12(Sv))) == P.fq12_flat(P.final_exponentiation(P.miller_loop([(P.affine_neg(P.F1, Pa), P.g2_prepare(Qa))])))
        out = G.simulate(progs["aggtail2"], {G.BUF_RAW3: P.fq12_flat(R), G.BUF_RAW2: Sv})
        assert (out == [1] + [0] * 11) is want


def test_cofactor_power_program_and_the_identity_it_rests_on(progs):
    """powc12raw: M -> M^(1 - x) for an arbitrary Fq12 element.  A large g2pubs VerifyAggregate pairs S_i = the hash point BEFORE its cofactor
    clearing (hash.go:306-309: H = [1 - x] S) and raises the product once: FE(ML(S, Q))^(1 - x) == FE(ML(H, Q)) for S on E(Fq) OUTSIDE the
    subgroup (the reduced pairing is bilinear in that argument modulo r E(Fq)), so FE(prod ML(S_i, Q_i)^(1 - x) * ML(-sig, G2)) == 1 is the reference's verdict."""
    xs = P.XORShift(31)
    c = P.BLS_X + 1
    f = P.miller_loop([(P.G1_GEN, P.g2_prepare(P.G2_GEN))])
    assert G.simulate(progs["powc12raw"], {G.BUF_RAW3: P.fq12_flat(f)}) == P.fq12_flat(P.fq12_pow(f, c))
    for trial in range(2):
        m = b"\x01" + b"cofactor-%d" % trial
        pp = P.jac_to_affine(P.F1, P.jac_add_affine(P.F1, P.to_jac(P.F1, P.swu_g1_helper(P.hp(m, 0))), P.swu_g1_helper(P.hp(m, 1))))
        S = P.iso11(pp)
        assert P.jac_to_affine(P.F1, P.affine_mul(P.F1, S, P.R_ORDER)) is not None        # S is NOT in the subgroup
        H = P.clear_h(S)
        assert H == P.hash_g1(m[1:])
        Qa = P.jac_to_affine(P.F2, P.affine_mul(P.F2, P.G2_GEN, P.rand_fr(xs)))
        prep = P.g2_prepare(Qa)
        ms = P.miller_loop([(S, prep)])
        powed = G.unflat12(G.simulate(progs["powc12raw"], {G.BUF_RAW3: P.fq12_flat(ms)}))
        assert P.final_exponentiation(powed) == P.final_exponentiation(P.miller_loop([(H, prep)]))


def _f2(v):
    return [v[0], v[1]]


def test_hash_tail_programs_match_oracle(progs):
    """the curve-arithmetic tails of HashG1 / HashG2 / HashG2WithDomain: outputs are the affine hash point, the check values
    are nonzero for ordinary inputs and the final one is zero when the result is the point at infinity"""
    msgs = [b"", b"abc", bytes(range(70))]
    for m in msgs:
        t1, t2 = P.hp(b"\x01" + m, 0), P.hp(b"\x01" + m, 1)
        p1, p2 = P.swu_g1_helper(t1), P.swu_g1_helper(t2)
        out = G.simulate(progs["hashfin1"], {0: [p1[0], p1[1], p2[0], p2[1]]})
        want = P.hash_g1(m)
        assert out[:2] == [want[0], want[1]] and all(out[2:])                    # hash.go:326-331
        u1, u2 = P.hp2(b"\x01" + m, 0), P.hp2(b"\x01" + m, 1)
        q1, q2 = P.swu_g2_helper(u1), P.swu_g2_helper(u2)
        out = G.simulate(progs["hashfin2"], {0: _f2(q1[0]) + _f2(q1[1]) + _f2(q2[0]) + _f2(q2[1])})
        want = P.hash_g2(m)
        assert out[:4] == _f2(want[0]) + _f2(want[1]) and all(out[4:])           # hash.go:405-411
    # opposite mapped points: the sum is the point at infinity, flagged by a zero check value (the caller falls back)
    p1 = P.swu_g1_helper(P.hp(b"\x01x", 0))
    out = G.simulate(progs["hashfin1"], {0: [p1[0], p1[1], p1[0], (-p1[1]) % P.Q]})
    assert out[2] == 0 and out[3] and out[4] and out[5] == 0
    # equal mapped points: flagged as well (the reference doubles with the a = 0 formula on the isogenous curve)
    out = G.simulate(progs["hashfin1"], {0: [p1[0], p1[1], p1[0], p1[1]]})
    assert out[5] == 0 and out[2] and out[3] and out[4]
    # ScaleByCofactor (g2.go:104-138) of the try-and-increment point
    msg32, dom = bytes(range(32)), bytes(range(8))
    x0 = (int.from_bytes(P._sha(msg32 + dom + b"\x01"), "big"), int.from_bytes(P._sha(msg32 + dom + b"\x02"), "big"))
    while True:
        y0 = P.fq2_sqrt(P.fq2_add(P.fq2_mul(P.fq2_sqr(x0), x0), P.B_COEFF_FQ2))
        if y0 is not None:
            break
        x0 = P.fq2_add(x0, P.FQ2_ONE)
    if not P.fq2_parity(y0):
        y0 = P.fq2_neg(y0)
    out = G.simulate(progs["cofac2"], {0: _f2(x0) + _f2(y0)})
    want = P.jac_to_affine(P.F2, P.hash_g2_with_domain(msg32, dom))
    assert out[:4] == _f2(want[0]) + _f2(want[1]) and out[4]


def test_subgroup_programs(progs):
    """[x^2] P + phi(P) / [|x|] P + psi(P): zero Z exactly for points of the prime-order subgroups (g1.go:137-141, g2.go:293-295)"""
    xs = P.XORShift(21)
    Pa, Qa = _pt(xs)
    assert G.simulate(progs["subgrp1"], {0: [Pa[0], Pa[1]]}) == [0]
    assert G.simulate(progs["subgrp2"], {0: [Qa[0][0], Qa[0][1], Qa[1][0], Qa[1][1]]}) == [0, 0]
    # curve points outside the subgroups (a random curve point is, with overwhelming probability)
    x = 5
    while True:
        y = P.fq_sqrt((x * x * x + 4) % P.Q)
        if y is not None:
            break
        x += 1
    assert not P.g1_in_subgroup((x, y))
    assert G.simulate(progs["subgrp1"], {0: [x, y]}) != [0]
    x2 = (7, 1)
    while True:
        y2 = P.fq2_sqrt(P.fq2_add(P.fq2_mul(P.fq2_sqr(x2), x2), P.B_COEFF_FQ2))
        if y2 is not None:
            break
        x2 = P.fq2_add(x2, P.FQ2_ONE)
    assert G.simulate(progs["subgrp2"], {0: [x2[0], x2[1], y2[0], y2[1]]}) != [0, 0]


def test_msm_final_programs():
    """the tail of the MSM over decomposed scalars: per window the fold's 2 + m sums (X, L, O_0 .. O_{m-1}) in Jacobian coordinates ->
    sum_w 2^(16 w) (L_w - X_w + 8 sum_l 2^l O_{w,l}), affine"""
    xs = P.XORShift(31)
    m, logk = 13, 3
    for kind, Fd, gen, six in (("msmfin1", P.F1, P.G1_GEN, False), ("msmfin2", P.F2, P.G2_GEN, True)):
        p = G.schedule(G.build_program(kind))
        nwin = 4 if six else 8
        inputs, total = {}, 0
        for w in range(nwin):
            ks = [P.rand_fr(xs) for _ in range(m + 2)]                          # X, L, O_0 .. O_{m-1} as multiples of the generator
            if w == 1:
                ks[0] = 0; ks[5] = 0                                            # empty sums: the point at infinity
            wt = (ks[1] - ks[0] + (1 << logk) * sum(ks[2 + l] << l for l in range(m))) % P.R_ORDER
            total = (total + (wt << (16 * w))) % P.R_ORDER
            for a, k in enumerate(ks):
                rec = a * nwin + w
                if k == 0:
                    coords = (0, 1, 0) if not six else ((0, 0), (1, 0), (0, 0))   # what the loader produces for a flagged record
                else:
                    pt = P.jac_to_affine(Fd, P.affine_mul(Fd, gen, k))
                    z = P.rand_int(xs, P.Q - 1) + 1                                # a random Jacobian representative (x z^2, y z^3, z)
                    if six:
                        zz = (z, 0); z2 = P.fq2_sqr(zz); z3 = P.fq2_mul(z2, zz)
                        coords = (P.fq2_mul(pt[0], z2), P.fq2_mul(pt[1], z3), zz)
                    else:
                        coords = (pt[0] * z * z % P.Q, pt[1] * z * z * z % P.Q, z)
                for j, cj in enumerate(coords):
                    if six:
                        inputs[G.soa_el(2 * j, rec, True)] = cj[0]; inputs[G.soa_el(2 * j + 1, rec, True)] = cj[1]
                    else:
                        inputs[G.soa_el(j, rec, False)] = cj
        out = G.simulate(p, {G.BUF_SOA3: inputs})
        want = P.jac_to_affine(Fd, P.affine_mul(Fd, gen, total % P.R_ORDER))
        if six:
            assert out[:4] == [want[0][0], want[0][1], want[1][0], want[1][1]] and out[4]
        else:
            assert out[:2] == [want[0], want[1]] and out[2]


def test_scalar_multiplication_programs():
    """[k] P with a run-time scalar (g1.go:80-90, g2.go MulFR) through the endomorphisms: SEL levels index the tables 0 P .. 15 P and
    their images under phi / psi by the 4-bit digits of the decomposed scalar"""
    import glv_model as GLV
    xs = P.XORShift(41)
    Pa, Qa = _pt(xs)
    for kind, Fd, pt, six in (("mul1", P.F1, Pa, False), ("mul2", P.F2, Qa, True)):
        p = G.schedule(G.build_program(kind))
        inp = [pt[0][0], pt[0][1], pt[1][0], pt[1][1]] if six else [pt[0], pt[1]]
        for k in (P.rand_fr(xs), 1, 0, P.R_ORDER - 1, (1 << 255) + 12345, P.R_ORDER, (1 << 256) - 1, GLV.Z2, GLV.Z**3 + 5):
            rec = (GLV.lat_record_g2 if six else GLV.lat_record_g1)(k)           # what k_glv_recode hands the program (glv_model.py)
            out = G.simulate(p, {0: inp, "scalar": rec})
            want = P.jac_to_affine(Fd, P.affine_mul(Fd, pt, k)) if k % P.R_ORDER else None
            if want is None:
                assert out[-1] == 0                                             # the point at infinity: Z = 0
            elif six:
                assert out[:4] == [want[0][0], want[0][1], want[1][0], want[1][1]] and out[4]
            else:
                assert out[:2] == [want[0], want[1]] and out[2]


def test_program_bounds_and_shape(progs):
    for name, p in progs.items():
        assert p.nslot < 1024
        for kind, jobs in p.levels:
            assert len(jobs) <= (1 if kind == G.K_INV else G.LANES)
            for n in jobs:
                if n.kind == "mul":
                    assert len(n.x) <= G.TMAX and len(n.y) <= G.TMAX
                    assert n.x.L() * n.y.L() <= G.LPROD_MAX and n.x.V() * n.y.V() <= G.VPROD_MAX
                    assert n.x.cmax() <= G.CMAX and n.y.cmax() <= G.CMAX
                elif n.kind == "lin":
                    assert len(n.x) <= G.TLIN and n.x.L() <= G.LMAX and n.x.cmax() <= G.CMAX
                for lin in (n.x, n.y):
                    if lin:
                        assert all(d.level < n.level for d in lin)                  # operands come from earlier levels
        blob = G.encode(p)
        assert len(blob) % 16 == 0


def test_product_tree_node_program():
    """mul12raw: two Fq12 values -> their product (fq12.go:198-213); a missing partner reads as 1"""
    p = G.schedule(G.build_program("mul12raw"))
    xs = P.XORShift(51)
    a = [P.rand_int(xs, P.Q) for _ in range(12)]; c = [P.rand_int(xs, P.Q) for _ in range(12)]
    inp = {e: a[e] for e in range(12)}; inp.update({e | 16: c[e] for e in range(12)})
    assert G.simulate(p, {G.BUF_SOA12: inp}) == P.fq12_flat(P.fq12_mul(G.unflat12(a), G.unflat12(c)))
    one = [1] + [0] * 11
    inp.update({e | 16: one[e] for e in range(12)})
    assert G.simulate(p, {G.BUF_SOA12: inp}) == a


def test_sum_tree_programs():
    """sum0 / sum1 / sumfin: one addition of the tree sum per program run (complete projective formulas; infinity as (0 : 1 : 0))"""
    xs = P.XORShift(61)
    Pa, Qa = _pt(xs); Pb, Qb = _pt(xs)
    for g, Fd, A, B, six in (("1", P.F1, Pa, Pb, False), ("2", P.F2, Qa, Qb, True)):
        bit = 32 if six else 0
        def proj(pt):                                                          # affine -> the elements a loader produces
            if pt is None:
                return [0, 0, 1, 0, 0, 0] if six else [0, 1, 0]
            return [pt[0][0], pt[0][1], pt[1][0], pt[1][1], 1, 0] if six else [pt[0], pt[1], 1]
        def run(kind, buf, X, Y):
            inp = {}
            for e, v in enumerate(proj(X)): inp[e | bit] = v
            for e, v in enumerate(proj(Y)): inp[e | 16 | bit] = v
            return G.simulate(G.schedule(G.build_program(kind)), {buf: inp})
        def aff(v):                                                            # projective elements -> affine tuple / None
            if six:
                z = (v[4], v[5])
                if z == (0, 0): return None
                zi = P.fq2_inv(z)
                return (P.fq2_mul((v[0], v[1]), zi), P.fq2_mul((v[2], v[3]), zi))
            if v[2] == 0: return None
            zi = pow(v[2], -1, P.Q)
            return (v[0] * zi % P.Q, v[1] * zi % P.Q)
        want = P.jac_to_affine(Fd, P.jac_add_affine(Fd, P.to_jac(Fd, A), B))
        s0 = run("sum0_" + g, G.BUF_AFFPT, A, B)
        assert aff(s0) == want
        assert aff(run("sum0_" + g, G.BUF_AFFPT, A, None)) == A and aff(run("sum0_" + g, G.BUF_AFFPT, None, None)) is None
        neg = (A[0], P.fq2_neg(A[1])) if six else (A[0], (-A[1]) % P.Q)
        assert aff(run("sum0_" + g, G.BUF_AFFPT, A, neg)) is None                   # P + (-P)
        dbl = P.jac_to_affine(Fd, P.jac_double(Fd, P.to_jac(Fd, A)))
        assert aff(run("sum0_" + g, G.BUF_AFFPT, A, A)) == dbl                      # P + P
        # inner level on projective records (use the level-0 output and an infinity partner), then the root
        inp = {e | bit: v for e, v in enumerate(s0)}
        inp.update({e | 16 | bit: v for e, v in enumerate(proj(None))})
        s1 = G.simulate(G.schedule(G.build_program("sum1_" + g)), {G.BUF_SOAPT: inp})
        assert aff(s1) == want
        fin = G.simulate(G.schedule(G.build_program("sumfin_" + g)), {G.BUF_SOAPT: {e | bit: v for e, v in enumerate(s1)}})
        flat = [want[0][0], want[0][1], want[1][0], want[1][1]] if six else [want[0], want[1]]
        assert fin[:len(flat)] == flat and fin[len(flat)]


def test_core_asm_blobs_are_current(tmp_path):
    """bls_amd/csrc/core_asm.inc (the multiply cores of the final-exponentiation kernels as assembly blobs) is generated from the
    compiler's own output for fp2p_mul_body / fp2p_sqr_body: regenerate it and compare, so that a change to the field arithmetic cannot
    leave a stale blob behind."""
    import shutil
    import subprocess
    import sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bls_amd", "csrc")
    out = tmp_path / "core_asm.inc"
    subprocess.check_call([sys.executable, os.path.join(csrc, "gen_core_asm.py"), str(tmp_path)], timeout=600)
    assert out.read_text() == open(os.path.join(csrc, "core_asm.inc")).read(), "core_asm.inc is stale: run python bls_amd/csrc/gen_core_asm.py"
    # ... and the blobs of the 14 x 28-bit build (k_fe_pair.hip with -DBLSMI_LIMBS28)
    out28 = tmp_path / "core_asm28.inc"
    subprocess.check_call([sys.executable, os.path.join(csrc, "gen_core_asm.py"), str(tmp_path), "--limbs28"], timeout=600)
    assert out28.read_text() == open(os.path.join(csrc, "core_asm28.inc")).read(), "core_asm28.inc is stale: run python bls_amd/csrc/gen_core_asm.py --limbs28"
