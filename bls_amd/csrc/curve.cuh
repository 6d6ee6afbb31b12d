// curve.cuh -- Jacobian group law on E: y^2 = x^3 + 4 (G1, over Fq) and E': y^2 = x^3 + 4(1+u)
// (G2, over Fq2), one point per lane.  Same formulas and special cases as the reference
// (g1.go:343-585, g2.go:389-632): dbl-2009-l, add-2007-bl, madd-2007-bl, MSB-first double-and-add.
// Infinity is z == 0 exactly as in the reference (g1.go:293, g2.go:331); because lazily reduced
// values have several representations of 0, an explicit all-ones/zero `inf` mask travels with every
// point instead of re-testing z.
#pragma once
#include "tower.cuh"

#define BLSMI_X_ABS 0xd201000000010000ULL          // |x| of the curve parameter; x < 0: blsIsNegative (g2.go:634-636)

namespace blsmi {

template <class F> struct Jac { F x, y, z; i32 inf; };    // F = FpS (G1) or Fp2S (G2); inf = -1 / 0
template <class F> struct Aff { F x, y; i32 inf; };
using G1Jac = Jac<FpS>; using G1Aff = Aff<FpS>;
using G2Jac = Jac<Fp2S>; using G2Aff = Aff<Fp2S>;

template <class F> BLSMI_DEV Jac<F> jac_zero() {                     // g1.go:275, g2.go:313: (0, 1, 0)
    Jac<F> p; p.x = field_consts<F>::zero(); p.y = field_consts<F>::one(); p.z = field_consts<F>::zero(); p.inf = -1; return p;
}
template <class F> BLSMI_DEV Jac<F> to_jac(const Aff<F>& a) {        // g1.go:59-64, g2.go:70-76
    Jac<F> p; p.x = a.x; p.y = a.y; p.z = field_consts<F>::one(); p.inf = a.inf;
    return p;
}
template <class F> BLSMI_DEV Jac<F> jac_select(i32 m, const Jac<F>& a, const Jac<F>& b) {
    Jac<F> r; r.x = f_select(m, a.x, b.x); r.y = f_select(m, a.y, b.y); r.z = f_select(m, a.z, b.z); r.inf = (a.inf & m) | (b.inf & ~m); return r;
}

// g1.go:343-397 / g2.go:389-443.  Doubling infinity returns infinity (z stays 0: nz = 2*y*z).
template <class F> BLSMI_DEV Jac<F> jac_double_i(const Jac<F>& g) {
    const F a = f_store(f_sqr(g.x));
    const F b = f_store(f_sqr(g.y));
    const F c = f_store(f_sqr(b));
    const F d = f_store(f_dbl(f_sub(f_sub(f_sqr(f_add(g.x, b)), a), c)));
    const F e = f_store(f_muls<3>(a));
    const auto f = f_sqr(e);
    Jac<F> r;
    r.z = f_store(f_dbl(f_mul(g.z, g.y)));
    r.x = f_store(f_sub(f_sub(f, d), d));
    r.y = f_store(f_sub(f_mul(f_sub(d, r.x), e), f_muls<8>(c)));
    r.inf = g.inf;
    return r;
}

// g1.go:485-559 / g2.go:532-606 (mixed addition).  Special cases as the reference:
// g infinite -> o; o infinite -> g; same point -> double; opposite points -> z3 = 0 (infinity).
template <class F> __device__ __noinline__ Jac<F> jac_double(const Jac<F>& g);
// the same, operand BY VALUE -- for the rare equal-points branch inside the inlined additions below: a reference parameter there would
// make the caller's accumulator an address-taken stack object, written to scratch on the HOT path as well (see xyzz_acc_affine_slow)
template <class F> __device__ __noinline__ Jac<F> jac_double_v(Jac<F> g);
template <class F> BLSMI_DEV Jac<F> jac_add_affine_i(const Jac<F>& g, const Aff<F>& o) {
    const F z1z1 = f_store(f_sqr(g.z));
    const F u2 = f_store(f_mul(o.x, z1z1));
    const F s2 = f_store(f_mul(f_mul(o.y, g.z), z1z1));
    const F h = f_store(f_sub(u2, g.x));
    const F rr0 = f_store(f_sub(s2, g.y));
    const bool h0 = f_is_zero(h);
    const bool live = (g.inf == 0) & (o.inf == 0);
    const F hh = f_store(f_sqr(h));
    const F i = f_store(f_muls<4>(hh));
    const F j = f_store(f_mul(h, i));
    const F rr = f_store(f_dbl(rr0));
    const F v = f_store(f_mul(g.x, i));
    Jac<F> r;
    r.x = f_store(f_sub(f_sub(f_sub(f_sqr(rr), j), v), v));
    r.y = f_store(f_sub(f_mul(f_sub(v, r.x), rr), f_dbl(f_mul(g.y, j))));
    r.z = f_store(f_sub(f_sub(f_sqr(f_add(g.z, h)), z1z1), hh));
    r.inf = 0;
    if (h0 & live) {                                                   // rare: same x
        if (f_is_zero(rr0)) r = jac_double_v(g);                         // same point (g1.go:506-509)
        else r.inf = -1;                                               // opposite points: z3 == 0
    }
    r = jac_select(g.inf, to_jac(o), r);                               // g1.go:486-488
    r = jac_select(o.inf & ~g.inf, g, r);                              // g1.go:489-491
    return r;
}

// g1.go:400-482 / g2.go:446-529 (general addition)
template <class F> BLSMI_DEV Jac<F> jac_add_i(const Jac<F>& g, const Jac<F>& o) {
    const F z1z1 = f_store(f_sqr(g.z));
    const F z2z2 = f_store(f_sqr(o.z));
    const F u1 = f_store(f_mul(g.x, z2z2));
    const F u2 = f_store(f_mul(o.x, z1z1));
    const F s1 = f_store(f_mul(f_mul(g.y, o.z), z2z2));
    const F s2 = f_store(f_mul(f_mul(o.y, g.z), z1z1));
    const F h = f_store(f_sub(u2, u1));
    const F rr0 = f_store(f_sub(s2, s1));
    const bool h0 = f_is_zero(h);
    const bool live = (g.inf == 0) & (o.inf == 0);
    const F i = f_store(f_sqr(f_dbl(h)));
    const F j = f_store(f_mul(h, i));
    const F rr = f_store(f_dbl(rr0));
    const F v = f_store(f_mul(u1, i));
    Jac<F> r;
    r.x = f_store(f_sub(f_sub(f_sub(f_sqr(rr), j), v), v));
    r.y = f_store(f_sub(f_mul(f_sub(v, r.x), rr), f_dbl(f_mul(s1, j))));
    r.z = f_store(f_mul(f_sub(f_sub(f_sqr(f_add(g.z, o.z)), z1z1), z2z2), h));
    r.inf = 0;
    if (h0 & live) {
        if (f_is_zero(rr0)) r = jac_double_v(g);
        else r.inf = -1;
    }
    r = jac_select(g.inf, o, r);
    r = jac_select(o.inf & ~g.inf, g, r);
    return r;
}

// Out-of-line forms (one copy of each body per translation unit): the point travels through the lane's scratch on every call, which is
// what the ladders with a point or two of live state want to avoid -- they use the *_i forms above and keep the accumulator in
// registers (rocprofv3, profiles/r03a: the out-of-line G1 kernels issued one VALU instruction per 8-9 cycles per SIMD against 5
// for the pairing kernels, behind 3.5 TB/s of argument traffic).
template <class F> __device__ __noinline__ Jac<F> jac_double(const Jac<F>& g) { return jac_double_i(g); }
template <class F> __device__ __noinline__ Jac<F> jac_double_v(Jac<F> g) { return jac_double_i(g); }
template <class F> __device__ __noinline__ Jac<F> jac_add_affine(const Jac<F>& g, const Aff<F>& o) { return jac_add_affine_i(g, o); }
// acc += o for loops that add MANY points into one accumulator (the MSM's bucket pass): the same mixed addition with its special cases
// (an infinite operand, equal x) BRANCHED out to the out-of-line function instead of selected at the end.  jac_add_affine_i keeps g, o
// and r alive to its last line for those selects -- 112 of a G1 lane's registers before any temporary, 50 spilled in the bucket kernel
// (6.3 GB of scratch written per 2^20-point MSM, rocprofv3 WRITE_SIZE) -- here every coordinate dies at its last use.  The rare path
// is taken by whole lanes only when one of their operands is special.
template <class F> __device__ __noinline__ Jac<F> jac_add_affine_v(Jac<F> g, Aff<F> o) { return jac_add_affine_i(g, o); }
template <class F> BLSMI_DEV void jac_acc_affine(Jac<F>& g, const Aff<F>& o) {
    const F z1z1 = f_store(f_sqr(g.z));
    const F h = f_store(f_sub(f_mul(o.x, z1z1), g.x));
    const F rr0 = f_store(f_sub(f_mul(f_mul(o.y, g.z), z1z1), g.y));
    if (__builtin_expect((g.inf != 0) | (o.inf != 0) | f_is_zero(h), 0)) { g = jac_add_affine_v(g, o); return; }
    const F hh = f_store(f_sqr(h));
    const F z3 = f_store(f_sub(f_sub(f_sqr(f_add(g.z, h)), z1z1), hh));
    const F i = f_store(f_muls<4>(hh));
    const F j = f_store(f_mul(h, i));
    const F v = f_store(f_mul(g.x, i));
    const F yj = f_store(f_dbl(f_mul(g.y, j)));
    const F rr = f_store(f_dbl(rr0));
    g.z = z3;
    g.x = f_store(f_sub(f_sub(f_sub(f_sqr(rr), j), v), v));
    g.y = f_store(f_sub(f_mul(f_sub(v, g.x), rr), yj));
}
template <class F> __device__ __noinline__ Jac<F> jac_add(const Jac<F>& g, const Jac<F>& o) { return jac_add_i(g, o); }

// The bucket accumulator of the MSM in XYZZ coordinates: (X, Y, ZZ, ZZZ) with x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2.  A bucket only ever
// ADDS affine points, and the mixed addition there is 8 multiplications + 2 squarings (madd-2008-s) where the Jacobian one is 7 + 4 and
// needs Z1^2, Z1^3 recomputed every time: 4 904 against 5 254 instructions per G1 addition (-7 %), -4.5 % for G2 (an Fq2 square costs 0.73
// of a product).  Same group element; the bucket leaves as the Jacobian triple (X ZZ, Y ZZZ, ZZ) -- two multiplications (xyzz_to_jac).  Special
// cases (an infinite operand, equal x) branch out as in jac_acc_affine, through the Jacobian out-of-line function.
template <class F> struct Xyzz { F x, y, zz, zzz; i32 inf; };
template <class F> BLSMI_DEV Xyzz<F> xyzz_zero() {
    Xyzz<F> p; p.x = field_consts<F>::zero(); p.y = field_consts<F>::one(); p.zz = field_consts<F>::zero(); p.zzz = field_consts<F>::zero(); p.inf = -1; return p;
}
template <class F> BLSMI_DEV Jac<F> xyzz_to_jac(const Xyzz<F>& g) {      // Z = ZZ: X' / Z^2 = X ZZ / ZZ^2 = x,  Y' / Z^3 = Y ZZZ / ZZ^3 = Y ZZZ / ZZZ^2 = y
    Jac<F> r;
    r.x = f_store(f_mul(g.x, g.zz));
    r.y = f_store(f_mul(g.y, g.zzz));
    r.z = g.zz;
    r.inf = g.inf;
    return r;
}
template <class F> BLSMI_DEV Xyzz<F> jac_to_xyzz(const Jac<F>& j) {
    Xyzz<F> r;
    r.x = j.x; r.y = j.y;
    r.zz = f_store(f_sqr(j.z));
    r.zzz = f_store(f_mul(r.zz, j.z));
    r.inf = j.inf;
    return r;
}
// (operands and result BY VALUE: a reference parameter of an out-of-line function makes the caller's accumulator an address-taken stack
//  object, and then it lives in scratch memory on the HOT path too -- the bucket pass of the MSM stored and re-loaded its 57-word
//  accumulator and the 29-word point around every addition: 6.3 GB of scratch writes per 2^20-point launch, profiles/r05a_counters.json)
template <class F> __device__ __noinline__ Xyzz<F> xyzz_acc_affine_slow(Xyzz<F> g, Aff<F> o) {
    if (g.inf != 0) { g.x = o.x; g.y = o.y; g.zz = field_consts<F>::one(); g.zzz = field_consts<F>::one(); g.inf = o.inf; return g; }   // (also o infinite: g stays infinite)
    if (o.inf != 0) return g;
    return jac_to_xyzz(jac_add_affine(xyzz_to_jac(g), o));                // equal x: the same point (doubling) or opposite points (infinity)
}
template <class F> BLSMI_DEV void xyzz_acc_affine(Xyzz<F>& g, const Aff<F>& o) {
    const F p = f_store(f_sub(f_mul(o.x, g.zz), g.x));                    // U2 - X1
    const F r = f_store(f_sub(f_mul(o.y, g.zzz), g.y));                   // S2 - Y1
    if (__builtin_expect((g.inf != 0) | (o.inf != 0) | f_is_zero(p), 0)) { g = xyzz_acc_affine_slow(g, o); return; }
    const F pp = f_store(f_sqr(p));
    const F ppp = f_store(f_mul(p, pp));
    const F q = f_store(f_mul(g.x, pp));
    const F yp = f_store(f_mul(g.y, ppp));
    g.zz = f_store(f_mul(g.zz, pp));
    g.zzz = f_store(f_mul(g.zzz, ppp));
    g.x = f_store(f_sub(f_sub(f_sub(f_sqr(r), ppp), q), q));
    g.y = f_store(f_sub(f_mul(f_sub(q, g.x), r), yp));
}

// g1.go:322-340 / g2.go:365-386
template <class F> __device__ __noinline__ Aff<F> jac_to_affine(const Jac<F>& g) {
    const F zi = f_inv(g.z);
    const F zi2 = f_store(f_sqr(zi));
    Aff<F> a;
    a.x = f_store(f_mul(g.x, zi2));
    a.y = f_store(f_mul(f_mul(g.y, zi2), zi));
    a.inf = g.inf;
    return a;
}
template <class F> BLSMI_DEV Jac<F> jac_neg(const Jac<F>& a) { Jac<F> r; r.x = a.x; r.y = f_store(f_neg(a.y)); r.z = a.z; r.inf = a.inf; return r; }
template <class F> BLSMI_DEV Aff<F> aff_neg(const Aff<F>& a) { Aff<F> r; r.x = a.x; r.y = f_store(f_neg(a.y)); r.inf = a.inf; return r; }

// MSB-first double-and-add over a 256-bit scalar held as 8 little-endian u32 words per lane
// (g1.go:67-90, g2.go:79-115).  The reference starts at BitLen(scalar); leading zero bits double the
// point at infinity, which is a no-op, so iterating all 256 bits gives the same point.
template <class F> BLSMI_DEV Jac<F> aff_mul_u256(const Aff<F>& p, const u32 k[8]) {
    Jac<F> res = jac_zero<F>();
    for (int i = 255; i >= 0; i--) {
        res = jac_double(res);
        const i32 bit = -(i32)((k[i >> 5] >> (i & 31)) & 1);
        const Jac<F> t = jac_add_affine(res, p);
        res = jac_select(bit, t, res);
    }
    return res;
}
// multiplication by a public 64-bit constant (ClearH, hash.go:306-309: |x| = 0xd201000000010000)
template <class F> BLSMI_DEV Jac<F> aff_mul_u64_public(const Aff<F>& p, u64 k) {
    Jac<F> res = to_jac(p);
#pragma unroll 1
    for (int i = 62 - __builtin_clzll(k); i >= 0; i--) {                  // the accumulator stays in registers (inlined bodies, one copy each)
        res = jac_double_i(res);
        if ((k >> i) & 1) res = jac_add_affine_i(res, p);
    }
    return res;
}
// the same for a Jacobian base point (general additions; g2.go:609-619)
template <class F> BLSMI_DEV Jac<F> jac_mul_u64_public(const Jac<F>& p, u64 k) {
    Jac<F> res = p;
#pragma unroll 1
    for (int i = 62 - __builtin_clzll(k); i >= 0; i--) {
        res = jac_double_i(res);
        if ((k >> i) & 1) res = jac_add_i(res, p);
    }
    return res;
}

// [d_0] T for the one digit of ScaleByCofactor (hash.cuh: scale_by_cofactor_g2), d_0 = (|x| + 1) / 3 = 0x4600 5555 5555 aaab: its
// sixteen nibbles are 0, 4, 5, 6, 10 or 11, so fixed 4-bit windows need five multiples of T (3 doublings + 3 additions to build) and
// 13 additions in the ladder -- 63 doublings + 16 additions against 62 + 27 bit by bit.
template <class F> BLSMI_DEV Jac<F> jac_mul_h2_d0(const Jac<F>& t) {
    static_assert(C_H2_D0 == 0x460055555555aaabull, "the nibble table below is that of this constant");
    const Jac<F> t2 = jac_double(t), t4 = jac_double(t2), t5 = jac_add(t4, t), t6 = jac_add(t5, t), t10 = jac_double(t5), t11 = jac_add(t10, t);
    Jac<F> tab[5];
    tab[0] = t4; tab[1] = t5; tab[2] = t6; tab[3] = t10; tab[4] = t11;
    Jac<F> res = t4;                                                       // the top nibble
#pragma unroll 1
    for (int i = 14; i >= 0; i--) {
#pragma unroll 1
        for (int d = 0; d < 4; d++) res = jac_double_i(res);
        const u32 nib = (u32)(C_H2_D0 >> (4 * i)) & 15u;
        if (nib) res = jac_add_i(res, tab[nib == 4 ? 0 : nib == 5 ? 1 : nib == 6 ? 2 : nib == 10 ? 3 : 4]);
    }
    return res;
}

}  // namespace blsmi
