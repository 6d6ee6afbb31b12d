/*
 * refcpu.c -- CPU ORACLE for the BLS12-381 verify path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the algorithms of phoreproject/bls (the Go reference cannot be built
 * in this image: no Go toolchain).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the timed CPU baseline --
 * never as the product path (bls_amd/ never links or dlopens it).
 *
 * Parity status: PINNED by the reference's own known-answer tests (tests/golden/reference_kats.json,
 * extracted from the reference's *_test.go files) and cross-checked against the independent
 * Python big-int twin oracle/pyref.py.
 *
 * Every function cites the reference file:line (under /root/reference) it follows.  The
 * algorithms -- separate 6x6 limb product + word-serial Montgomery reduction, Karatsuba towers,
 * bit-serial scalar multiplication, 68-step prepared Miller loop, generic (non-cyclotomic)
 * FQ12.Exp in the final exponentiation, one full pairing per message in VerifyAggregate -- are the
 * reference's, so the timing of this code is a fair "C port of the reference algorithm" baseline.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef uint8_t u8;

typedef struct { u64 l[6]; } fq;      /* Montgomery form, R = 2^384, always in [0,q) (fq.go:41-45) */
typedef struct { fq c0, c1; } fq2;
typedef struct { fq2 c0, c1, c2; } fq6;
typedef struct { fq6 c0, c1; } fq12;

#include "refcpu_consts.h"

#define API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * L0: limb primitives (stub_fallback.go:11-155)
 * ---------------------------------------------------------------------------------------- */
static inline u64 mac(u64 a, u64 b, u64 c, u64 *carry) {          /* stub_fallback.go:149-155 */
    u128 t = (u128)b * c + a + *carry;
    *carry = (u64)(t >> 64);
    return (u64)t;
}
static inline u64 adc(u64 a, u64 b, u64 *carry) {                 /* stub_fallback.go:135-139 */
    u128 t = (u128)a + b + *carry;
    *carry = (u64)(t >> 64);
    return (u64)t;
}
static inline u64 sbb(u64 a, u64 b, u64 *borrow) {                /* stub_fallback.go:143-146 */
    u128 t = (u128)a - b - *borrow;
    *borrow = (u64)(t >> 127);
    return (u64)t;
}
API void rc_mac_with_carry(u64 a, u64 b, u64 c, u64 carry, u64 *out, u64 *ncarry) { *ncarry = carry; *out = mac(a, b, c, ncarry); }
API void rc_add_with_carry(u64 a, u64 b, u64 carry, u64 *out, u64 *ncarry) { *ncarry = carry; *out = adc(a, b, ncarry); }
API void rc_sub_with_borrow(u64 a, u64 b, u64 borrow, u64 *out, u64 *nborrow) { *nborrow = borrow; *out = sbb(a, b, nborrow); }

/* stub_fallback.go:11-57: 6x6 schoolbook product, operand scanning */
API void rc_multiply_fqrepr(const u64 a[6], const u64 b[6], u64 hi[6], u64 lo[6]) {
    u64 t[12];
    memset(t, 0, sizeof t);
    for (int i = 0; i < 6; i++) {
        u64 carry = 0;
        for (int j = 0; j < 6; j++) t[i + j] = mac(t[i + j], a[i], b[j], &carry);
        t[i + 6] = carry;
    }
    memcpy(lo, t, 48);
    memcpy(hi, t + 6, 48);
}

/* stub_fallback.go:61-116: word-serial Montgomery reduction; result in [0,2q) */
API void rc_mont_reduce(const u64 hi_in[6], const u64 lo_in[6], u64 out[6]) {
    u64 t[12];
    memcpy(t, lo_in, 48);
    memcpy(t + 6, hi_in, 48);
    u64 carry2 = 0;
    for (int i = 0; i < 6; i++) {
        u64 k = t[i] * RC_QINV, carry = 0;
        (void)mac(t[i], k, RC_Q[0], &carry);
        for (int j = 1; j < 6; j++) t[i + j] = mac(t[i + j], k, RC_Q[j], &carry);
        u128 s = (u128)t[i + 6] + carry2 + carry;                  /* AddWithCarry(hi[i], carry2, carry) */
        t[i + 6] = (u64)s;
        carry2 = (u64)(s >> 64);
    }
    memcpy(out, t + 6, 48);
}

/* ------------------------------------------------------------------------------------------
 * L1: FQRepr helpers (fqrepr.go)
 * ---------------------------------------------------------------------------------------- */
static int repr_cmp(const u64 *a, const u64 *b, int n) {            /* fqrepr.go:143-155 */
    for (int i = n - 1; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return -1; }
    return 0;
}
static int repr_is_zero(const u64 *a, int n) { for (int i = 0; i < n; i++) if (a[i]) return 0; return 1; }
static void repr_add(u64 *a, const u64 *b) { u64 c = 0; for (int i = 0; i < 6; i++) a[i] = adc(a[i], b[i], &c); }   /* AddNoCarry */
static void repr_sub(u64 *a, const u64 *b) { u64 c = 0; for (int i = 0; i < 6; i++) a[i] = sbb(a[i], b[i], &c); }   /* SubNoBorrow */
static void repr_div2(u64 *a) { u64 t = 0; for (int i = 5; i >= 0; i--) { u64 t2 = a[i] << 63; a[i] = (a[i] >> 1) | t; t = t2; } }  /* fqrepr.go:89-97 */
static void repr_mul2(u64 *a) { u64 last = 0; for (int i = 0; i < 6; i++) { u64 tmp = a[i] >> 63; a[i] = (a[i] << 1) | last; last = tmp; } }
static unsigned repr_bitlen(const u64 *a, int n) {                  /* fqrepr.go:168-180 */
    for (int i = n - 1; i >= 0; i--) if (a[i]) return 64 * i + 64 - __builtin_clzll(a[i]);
    return 0;
}
static int repr_bit(const u64 *a, unsigned n) { return (a[n / 64] >> (n % 64)) & 1; }
static void repr_from_be(u64 *out, const u8 *b, int nlimbs) {       /* fqrepr.go:182-190 */
    for (int i = 0; i < nlimbs; i++) { u64 v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | b[8 * (nlimbs - 1 - i) + k]; out[i] = v; }
}
static void repr_to_be(u8 *b, const u64 *in, int nlimbs) {          /* fqrepr.go:193-202 */
    for (int i = 0; i < nlimbs; i++) for (int k = 0; k < 8; k++) b[8 * (nlimbs - 1 - i) + k] = (u8)(in[i] >> (56 - 8 * k));
}

/* ------------------------------------------------------------------------------------------
 * L2: Fq (fq.go)
 * ---------------------------------------------------------------------------------------- */
static const fq FQ_ZERO = {{0, 0, 0, 0, 0, 0}};
static inline int fq_is_valid(const fq *a) { return (a->l[5] & 0xf000000000000000ULL) == 0 || repr_cmp(a->l, RC_Q, 6) < 0; }   /* fq.go:37-39 */
static inline void fq_reduce(fq *a) { if (!fq_is_valid(a)) repr_sub(a->l, RC_Q); }                                               /* fq.go:41-45 */
static inline int fq_is_zero(const fq *a) { return repr_is_zero(a->l, 6); }
static inline int fq_eq(const fq *a, const fq *b) { return memcmp(a, b, sizeof(fq)) == 0; }
static void fq_add(fq *r, const fq *a, const fq *b) { fq t = *a; repr_add(t.l, b->l); fq_reduce(&t); *r = t; }                    /* fq.go:64-67 */
static void fq_sub(fq *r, const fq *a, const fq *b) {                                                                              /* fq.go:82-87 */
    fq t = *a;
    if (repr_cmp(b->l, t.l, 6) > 0) repr_add(t.l, RC_Q);
    repr_sub(t.l, b->l);
    *r = t;
}
static void fq_neg(fq *r, const fq *a) { if (fq_is_zero(a)) { *r = *a; return; } fq t; memcpy(t.l, RC_Q, 48); repr_sub(t.l, a->l); *r = t; }  /* fq.go:121-127 */
static void fq_dbl(fq *r, const fq *a) { fq t = *a; repr_mul2(t.l); fq_reduce(&t); *r = t; }                                    /* fq.go:140-143 */
static void fq_mul(fq *r, const fq *a, const fq *b) {                                                                              /* fq.go:70-79 */
    u64 hi[6], lo[6];
    rc_multiply_fqrepr(a->l, b->l, hi, lo);
    rc_mont_reduce(hi, lo, r->l);
    fq_reduce(r);
}
static void fq_sqr(fq *r, const fq *a) {                                                                                           /* fq.go:151-198 */
    /* off-diagonal products, doubled, plus diagonal -- same value as the reference's dedicated routine */
    const u64 *n = a->l;
    u64 t[12];
    memset(t, 0, sizeof t);
    for (int i = 0; i < 5; i++) {
        u64 carry = 0;
        for (int j = i + 1; j < 6; j++) t[i + j] = mac(t[i + j], n[i], n[j], &carry);
        t[i + 6] = carry;
    }
    t[11] = t[10] >> 63;
    for (int i = 10; i >= 2; i--) t[i] = (t[i] << 1) | (t[i - 1] >> 63);
    t[1] <<= 1;
    u64 carry = 0;
    for (int i = 0; i < 6; i++) {
        t[2 * i] = mac(t[2 * i], n[i], n[i], &carry);
        t[2 * i + 1] = adc(t[2 * i + 1], 0, &carry);
    }
    rc_mont_reduce(t + 6, t, r->l);
    fq_reduce(r);
}
static void fq_from_repr(fq *r, const u64 repr[6]) {                                                                               /* fq.go:49-56: invalid -> 0 */
    fq t; memcpy(t.l, repr, 48);
    if (!fq_is_valid(&t)) { *r = FQ_ZERO; return; }
    fq r2; memcpy(r2.l, RC_R2, 48);
    fq_mul(r, &t, &r2);
}
static void fq_to_repr(u64 out[6], const fq *a) {                                                                                  /* fq.go:334-338 */
    u64 z[6] = {0, 0, 0, 0, 0, 0};
    fq t;
    rc_mont_reduce(z, a->l, t.l);
    fq_reduce(&t);
    memcpy(out, t.l, 48);
}
static void fq_exp(fq *r, const fq *a, const u64 *e, int nlimbs) {                                                                 /* fq.go:96-113 (MSB first) */
    fq res = RC_ONE, base = *a;
    int found = 0;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        int bit = repr_bit(e, i);
        if (found) fq_sqr(&res, &res); else found = bit;
        if (bit) fq_mul(&res, &res, &base);
    }
    *r = res;
}
static int fq_inverse(fq *r, const fq *a) {                                                                                        /* fq.go:224-266 */
    if (fq_is_zero(a)) return 0;
    u64 u[6], v[6], one[6] = {1, 0, 0, 0, 0, 0};
    memcpy(u, a->l, 48);
    memcpy(v, RC_Q, 48);
    fq b, c = FQ_ZERO;
    memcpy(b.l, RC_R2, 48);
    while (repr_cmp(u, one, 6) != 0 && repr_cmp(v, one, 6) != 0) {
        while ((u[0] & 1) == 0) {
            repr_div2(u);
            if ((b.l[0] & 1) == 0) repr_div2(b.l); else { repr_add(b.l, RC_Q); repr_div2(b.l); }
        }
        while ((v[0] & 1) == 0) {
            repr_div2(v);
            if ((c.l[0] & 1) == 0) repr_div2(c.l); else { repr_add(c.l, RC_Q); repr_div2(c.l); }
        }
        if (repr_cmp(u, v, 6) >= 0) { repr_sub(u, v); fq_sub(&b, &b, &c); }
        else { repr_sub(v, u); fq_sub(&c, &c, &b); }
    }
    *r = (repr_cmp(u, one, 6) == 0) ? b : c;
    return 1;
}
static int fq_sqrt(fq *r, const fq *a) {                                                                                           /* fq.go:203-217 */
    fq a1, a0;
    fq_exp(&a1, a, RC_QM3O4, 6);
    fq_sqr(&a0, &a1);
    fq_mul(&a0, &a0, a);
    if (fq_eq(&a0, &RC_NEGONE)) return 0;
    fq_mul(r, &a1, a);
    return 1;
}
static int fq_cmp(const fq *a, const fq *b) { u64 x[6], y[6]; fq_to_repr(x, a); fq_to_repr(y, b); return repr_cmp(x, y, 6); }      /* fq.go:134-137 */
static int fq_parity(const fq *a) { fq n; fq_neg(&n, a); return fq_cmp(a, &n) > 0; }                                              /* fq.go:269-273 */

/* ------------------------------------------------------------------------------------------
 * L3: Fq2 (fq2.go)
 * ---------------------------------------------------------------------------------------- */
static const fq2 FQ2_ZERO = {{{0}}, {{0}}};
static fq2 fq2_one(void) { fq2 o; o.c0 = RC_ONE; o.c1 = FQ_ZERO; return o; }
static int fq2_is_zero(const fq2 *a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static int fq2_eq(const fq2 *a, const fq2 *b) { return memcmp(a, b, sizeof(fq2)) == 0; }
static void fq2_add(fq2 *r, const fq2 *a, const fq2 *b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static void fq2_sub(fq2 *r, const fq2 *a, const fq2 *b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
static void fq2_neg(fq2 *r, const fq2 *a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
static void fq2_dbl(fq2 *r, const fq2 *a) { fq_dbl(&r->c0, &a->c0); fq_dbl(&r->c1, &a->c1); }
static void fq2_mul_nr(fq2 *r, const fq2 *a) { fq t0 = a->c0; fq_sub(&r->c0, &a->c0, &a->c1); fq_add(&r->c1, &a->c1, &t0); }    /* fq2.go:41-45 */
static void fq2_mul(fq2 *r, const fq2 *a, const fq2 *b) {                                                                          /* fq2.go:116-130 */
    fq aa, bb, o, c1;
    fq_mul(&aa, &a->c0, &b->c0);
    fq_mul(&bb, &a->c1, &b->c1);
    fq_add(&o, &b->c0, &b->c1);
    fq_add(&c1, &a->c1, &a->c0);
    fq_mul(&c1, &c1, &o);
    fq_sub(&c1, &c1, &aa);
    fq_sub(&c1, &c1, &bb);
    fq_sub(&r->c0, &aa, &bb);
    r->c1 = c1;
}
static void fq2_sqr(fq2 *r, const fq2 *a) {                                                                                        /* fq2.go:75-89 */
    fq ab, c0c1, c0;
    fq_mul(&ab, &a->c0, &a->c1);
    fq_add(&c0c1, &a->c0, &a->c1);
    fq_neg(&c0, &a->c1);
    fq_add(&c0, &c0, &a->c0);
    fq_mul(&c0, &c0, &c0c1);
    fq_sub(&c0, &c0, &ab);
    fq_add(&c0, &c0, &ab);
    fq_add(&ab, &ab, &ab);
    r->c0 = c0; r->c1 = ab;
}
static void fq2_mul_fq(fq2 *r, const fq2 *a, const fq *s) { fq_mul(&r->c0, &a->c0, s); fq_mul(&r->c1, &a->c1, s); }
static int fq2_inverse(fq2 *r, const fq2 *a) {                                                                                     /* fq2.go:133-147 */
    fq t0, t1, t;
    fq_sqr(&t1, &a->c1);
    fq_sqr(&t0, &a->c0);
    fq_add(&t0, &t0, &t1);
    if (!fq_inverse(&t, &t0)) return 0;
    fq_mul(&r->c0, &a->c0, &t);
    fq_mul(&r->c1, &a->c1, &t);
    fq_neg(&r->c1, &r->c1);
    return 1;
}
static void fq2_frob(fq2 *r, const fq2 *a, unsigned power) {                                                                       /* fq2.go:156-158 */
    *r = *a;
    if (power % 2) fq_mul(&r->c1, &r->c1, &RC_NEGONE); else fq_mul(&r->c1, &r->c1, &RC_ONE);
}
static int fq2_cmp(const fq2 *a, const fq2 *b) { int c = fq_cmp(&a->c1, &b->c1); return c ? c : fq_cmp(&a->c0, &b->c0); }        /* fq2.go:31-37 */
static int fq2_parity(const fq2 *a) { fq2 n; fq2_neg(&n, a); return fq2_cmp(a, &n) > 0; }                                        /* fq2.go:256-260 */
static void fq2_exp(fq2 *r, const fq2 *a, const u64 *e, int nlimbs) {                                                              /* fq2.go:177-193 */
    fq2 res = fq2_one(), base = *a;
    int found = 0;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        int bit = repr_bit(e, i);
        if (found) fq2_sqr(&res, &res); else found = bit;
        if (bit) fq2_mul(&res, &res, &base);
    }
    *r = res;
}
static int fq2_sqrt(fq2 *r, const fq2 *a) {                                                                                        /* fq2.go:198-232 */
    if (fq2_is_zero(a)) { *r = FQ2_ZERO; return 1; }
    fq2 a1, alpha, a0, neg1;
    fq2_exp(&a1, a, RC_QM3O4, 6);
    fq2_sqr(&alpha, &a1);
    fq2_mul(&alpha, &alpha, a);
    fq2_frob(&a0, &alpha, 1);
    fq2_mul(&a0, &a0, &alpha);
    neg1.c0 = RC_NEGONE; neg1.c1 = FQ_ZERO;
    if (fq2_eq(&a0, &neg1)) return 0;
    fq2_mul(&a1, &a1, a);
    if (fq2_eq(&alpha, &neg1)) {
        fq2 u; u.c0 = FQ_ZERO; u.c1 = RC_ONE;
        fq2_mul(r, &a1, &u);
        return 1;
    }
    fq2 one = fq2_one();
    fq2_add(&alpha, &alpha, &one);
    fq2_exp(&alpha, &alpha, RC_QM1O2, 6);
    fq2_mul(r, &alpha, &a1);
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * L3: Fq6 (fq6.go)
 * ---------------------------------------------------------------------------------------- */
static void fq6_add(fq6 *r, const fq6 *a, const fq6 *b) { fq2_add(&r->c0, &a->c0, &b->c0); fq2_add(&r->c1, &a->c1, &b->c1); fq2_add(&r->c2, &a->c2, &b->c2); }
static void fq6_sub(fq6 *r, const fq6 *a, const fq6 *b) { fq2_sub(&r->c0, &a->c0, &b->c0); fq2_sub(&r->c1, &a->c1, &b->c1); fq2_sub(&r->c2, &a->c2, &b->c2); }
static void fq6_neg(fq6 *r, const fq6 *a) { fq2_neg(&r->c0, &a->c0); fq2_neg(&r->c1, &a->c1); fq2_neg(&r->c2, &a->c2); }
static void fq6_mul_nr(fq6 *r, const fq6 *a) { fq6 t; fq2_mul_nr(&t.c0, &a->c2); t.c1 = a->c0; t.c2 = a->c1; *r = t; }           /* fq6.go:34-37 */
static void fq6_mul(fq6 *r, const fq6 *a, const fq6 *b) {                                                                          /* fq6.go:255-292 */
    fq2 aa, bb, cc, tmp, t1, t2, t3;
    fq2_mul(&aa, &a->c0, &b->c0);
    fq2_mul(&bb, &a->c1, &b->c1);
    fq2_mul(&cc, &a->c2, &b->c2);
    fq2_add(&tmp, &a->c1, &a->c2);
    fq2_add(&t1, &b->c1, &b->c2);
    fq2_mul(&t1, &t1, &tmp);
    fq2_sub(&t1, &t1, &bb);
    fq2_sub(&t1, &t1, &cc);
    fq2_mul_nr(&t1, &t1);
    fq2_add(&t1, &t1, &aa);
    fq2_add(&tmp, &a->c0, &a->c2);
    fq2_add(&t3, &b->c0, &b->c2);
    fq2_mul(&t3, &t3, &tmp);
    fq2_sub(&t3, &t3, &aa);
    fq2_add(&t3, &t3, &bb);
    fq2_sub(&t3, &t3, &cc);
    fq2_add(&tmp, &a->c0, &a->c1);
    fq2_add(&t2, &b->c0, &b->c1);
    fq2_mul(&t2, &t2, &tmp);
    fq2_sub(&t2, &t2, &aa);
    fq2_sub(&t2, &t2, &bb);
    fq2_mul_nr(&cc, &cc);
    fq2_add(&t2, &t2, &cc);
    r->c0 = t1; r->c1 = t2; r->c2 = t3;
}
static void fq6_sqr(fq6 *r, const fq6 *a) {                                                                                        /* fq6.go:221-252 */
    fq2 s0, ab, s1, s2, bc, s3, s4, c0, c1, c2;
    fq2_sqr(&s0, &a->c0);
    fq2_mul(&ab, &a->c0, &a->c1);
    fq2_dbl(&s1, &ab);
    fq2_sub(&s2, &a->c0, &a->c1);
    fq2_add(&s2, &s2, &a->c2);
    fq2_sqr(&s2, &s2);
    fq2_mul(&bc, &a->c1, &a->c2);
    fq2_dbl(&s3, &bc);
    fq2_sqr(&s4, &a->c2);
    fq2_mul_nr(&c0, &s3); fq2_add(&c0, &c0, &s0);
    fq2_mul_nr(&c1, &s4); fq2_add(&c1, &c1, &s1);
    fq2_add(&c2, &s1, &s2); fq2_add(&c2, &c2, &s3); fq2_sub(&c2, &c2, &s0); fq2_sub(&c2, &c2, &s4);
    r->c0 = c0; r->c1 = c1; r->c2 = c2;
}
static void fq6_mul_by_1(fq6 *r, const fq6 *a, const fq2 *c1) {                                                                    /* fq6.go:40-57 */
    fq2 b, tmp, t1, t2;
    fq2_mul(&b, &a->c1, c1);
    fq2_add(&tmp, &a->c1, &a->c2);
    fq2_mul(&t1, c1, &tmp); fq2_sub(&t1, &t1, &b); fq2_mul_nr(&t1, &t1);
    fq2_add(&tmp, &a->c0, &a->c1);
    fq2_mul(&t2, c1, &tmp); fq2_sub(&t2, &t2, &b);
    r->c0 = t1; r->c1 = t2; r->c2 = b;
}
static void fq6_mul_by_01(fq6 *r, const fq6 *a, const fq2 *c0, const fq2 *c1) {                                                    /* fq6.go:60-90 */
    fq2 aa, b, tmp, t1, t2, t3;
    fq2_mul(&aa, &a->c0, c0);
    fq2_mul(&b, &a->c1, c1);
    fq2_add(&tmp, &a->c1, &a->c2);
    fq2_mul(&t1, c1, &tmp); fq2_sub(&t1, &t1, &b); fq2_mul_nr(&t1, &t1); fq2_add(&t1, &t1, &aa);
    fq2_add(&tmp, &a->c0, &a->c2);
    fq2_mul(&t3, c0, &tmp); fq2_sub(&t3, &t3, &aa); fq2_add(&t3, &t3, &b);
    fq2_add(&tmp, &a->c0, &a->c1);
    fq2_add(&t2, c0, c1); fq2_mul(&t2, &t2, &tmp); fq2_sub(&t2, &t2, &aa); fq2_sub(&t2, &t2, &b);
    r->c0 = t1; r->c1 = t2; r->c2 = t3;
}
static int fq6_inverse(fq6 *r, const fq6 *a) {                                                                                     /* fq6.go:295-336 */
    fq2 c0, c0s, c1, c0c1, c0c2, c2, tmp1, tmp2;
    fq2_mul_nr(&c0, &a->c2); fq2_mul(&c0, &c0, &a->c1); fq2_neg(&c0, &c0);
    fq2_sqr(&c0s, &a->c0); fq2_add(&c0, &c0, &c0s);
    fq2_sqr(&c1, &a->c2); fq2_mul_nr(&c1, &c1);
    fq2_mul(&c0c1, &a->c0, &a->c1);
    fq2_mul(&c0c2, &a->c0, &a->c2);
    fq2_sub(&c1, &c1, &c0c1);
    fq2_sqr(&c2, &a->c1); fq2_sub(&c2, &c2, &c0c2);
    fq2_mul(&tmp1, &a->c2, &c1);
    fq2_mul(&tmp2, &a->c1, &c2);
    fq2_add(&tmp1, &tmp1, &tmp2); fq2_mul_nr(&tmp1, &tmp1);
    fq2_mul(&tmp2, &a->c0, &c0);
    fq2_add(&tmp1, &tmp1, &tmp2);
    if (!fq2_inverse(&tmp1, &tmp1)) return 0;
    fq2_mul(&r->c0, &tmp1, &c0); fq2_mul(&r->c1, &tmp1, &c1); fq2_mul(&r->c2, &tmp1, &c2);
    return 1;
}
static void fq6_frob(fq6 *r, const fq6 *a, unsigned power) {                                                                       /* fq6.go:211-218 */
    fq6 t;
    fq2_frob(&t.c0, &a->c0, power); fq2_frob(&t.c1, &a->c1, power); fq2_frob(&t.c2, &a->c2, power);
    fq2_mul(&t.c1, &t.c1, &RC_FROB6_C1[power % 6]);
    fq2_mul(&t.c2, &t.c2, &RC_FROB6_C2[power % 6]);
    *r = t;
}

/* ------------------------------------------------------------------------------------------
 * L3: Fq12 (fq12.go)
 * ---------------------------------------------------------------------------------------- */
static fq12 fq12_one(void) { fq12 o; memset(&o, 0, sizeof o); o.c0.c0.c0 = RC_ONE; return o; }
static int fq12_eq(const fq12 *a, const fq12 *b) { return memcmp(a, b, sizeof(fq12)) == 0; }
static void fq12_conj(fq12 *r, const fq12 *a) { r->c0 = a->c0; fq6_neg(&r->c1, &a->c1); }                                        /* fq12.go:27-29 */
static void fq12_mul(fq12 *r, const fq12 *a, const fq12 *b) {                                                                      /* fq12.go:198-213 */
    fq6 aa, bb, o, c1, c0;
    fq6_mul(&aa, &a->c0, &b->c0);
    fq6_mul(&bb, &a->c1, &b->c1);
    fq6_add(&o, &b->c0, &b->c1);
    fq6_add(&c1, &a->c1, &a->c0);
    fq6_mul(&c1, &c1, &o);
    fq6_sub(&c1, &c1, &aa);
    fq6_sub(&c1, &c1, &bb);
    fq6_mul_nr(&c0, &bb);
    fq6_add(&c0, &c0, &aa);
    r->c0 = c0; r->c1 = c1;
}
static void fq12_sqr(fq12 *r, const fq12 *a) {                                                                                     /* fq12.go:180-195 */
    fq6 ab, c0c1, c0, c1;
    fq6_mul(&ab, &a->c0, &a->c1);
    fq6_add(&c0c1, &a->c0, &a->c1);
    fq6_mul_nr(&c0, &a->c1);
    fq6_add(&c0, &c0, &a->c0);
    fq6_mul(&c0, &c0, &c0c1);
    fq6_sub(&c0, &c0, &ab);
    fq6_add(&c1, &ab, &ab);
    fq6_mul_nr(&ab, &ab);
    fq6_sub(&c0, &c0, &ab);
    r->c0 = c0; r->c1 = c1;
}
static void fq12_mul_by_014(fq12 *r, const fq12 *a, const fq2 *c0, const fq2 *c1, const fq2 *c4) {                                /* fq12.go:32-47 */
    fq6 aa, bb, t;
    fq2 o;
    fq6_mul_by_01(&aa, &a->c0, c0, c1);
    fq6_mul_by_1(&bb, &a->c1, c4);
    fq2_add(&o, c1, c4);
    fq6_add(&t, &a->c1, &a->c0);
    fq6_mul_by_01(&t, &t, c0, &o);
    fq6_sub(&t, &t, &aa);
    fq6_sub(&t, &t, &bb);
    fq6 n; fq6_mul_nr(&n, &bb);
    fq6_add(&r->c0, &n, &aa);
    r->c1 = t;
}
static int fq12_inverse(fq12 *r, const fq12 *a) {                                                                                  /* fq12.go:216-237 */
    fq6 c0s, c1s;
    fq6_sqr(&c0s, &a->c0);
    fq6_sqr(&c1s, &a->c1);
    fq6_mul_nr(&c1s, &c1s);
    fq6_sub(&c0s, &c0s, &c1s);
    if (!fq6_inverse(&c0s, &c0s)) return 0;
    fq6 t0, t1;
    fq6_mul(&t0, &c0s, &a->c0);
    fq6_mul(&t1, &c0s, &a->c1);
    fq6_neg(&t1, &t1);
    r->c0 = t0; r->c1 = t1;
    return 1;
}
static void fq12_frob(fq12 *r, const fq12 *a, unsigned power) {                                                                    /* fq12.go:171-177 */
    fq12 t;
    fq6_frob(&t.c0, &a->c0, power);
    fq6_frob(&t.c1, &a->c1, power);
    fq2_mul(&t.c1.c0, &t.c1.c0, &RC_FROB12_C1[power % 12]);
    fq2_mul(&t.c1.c1, &t.c1.c1, &RC_FROB12_C1[power % 12]);
    fq2_mul(&t.c1.c2, &t.c1.c2, &RC_FROB12_C1[power % 12]);
    *r = t;
}
static void fq12_exp_u64(fq12 *r, const fq12 *a, u64 e) {                                                                          /* fq12.go:108-120: LSB first, full multiplications */
    fq12 res = fq12_one(), fi = *a;
    while (e) {
        if (e & 1) fq12_mul(&res, &res, &fi);
        fq12_mul(&fi, &fi, &fi);
        e >>= 1;
    }
    *r = res;
}

/* ------------------------------------------------------------------------------------------
 * L4: curve groups -- the same Jacobian formulas over Fq (G1) and Fq2 (G2)  (g1.go, g2.go)
 * ---------------------------------------------------------------------------------------- */
static fq fq_one_v(void) { return RC_ONE; }
#define FE fq
#define FN(n) fq_##n
#define GN(n) g1_##n
#define F_ONE fq_one_v()
#define F_ZERO FQ_ZERO
#include "refcpu_curve.inc"
#undef FE
#undef FN
#undef GN
#undef F_ONE
#undef F_ZERO
#define FE fq2
#define FN(n) fq2_##n
#define GN(n) g2_##n
#define F_ONE fq2_one()
#define F_ZERO FQ2_ZERO
#include "refcpu_curve.inc"
#undef FE
#undef FN
#undef GN
#undef F_ONE
#undef F_ZERO

static g1_aff g1_generator(void) { g1_aff g; g.x = RC_G1X; g.y = RC_G1Y; g.inf = 0; return g; }
static g2_aff g2_generator(void) { g2_aff g; g.x = RC_G2X; g.y = RC_G2Y; g.inf = 0; return g; }

/* ------------------------------------------------------------------------------------------
 * G2 prepare (g2.go:634-801)
 * ---------------------------------------------------------------------------------------- */
#define BLS_X 0xd201000000010000ULL
#define N_COEFFS 68
typedef struct { fq2 c[N_COEFFS][3]; int inf; } g2_prepared;

static void doubling_step(g2_jac *r, fq2 out[3]) {                                                                                 /* g2.go:655-708 */
    fq2 tmp0, tmp1, tmp2, tmp3, tmp4, tmp5, tmp6, zsq;
    fq2_sqr(&tmp0, &r->x);
    fq2_sqr(&tmp1, &r->y);
    fq2_sqr(&tmp2, &tmp1);
    fq2_add(&tmp3, &tmp1, &r->x); fq2_sqr(&tmp3, &tmp3); fq2_sub(&tmp3, &tmp3, &tmp0); fq2_sub(&tmp3, &tmp3, &tmp2); fq2_dbl(&tmp3, &tmp3);
    fq2_dbl(&tmp4, &tmp0); fq2_add(&tmp4, &tmp4, &tmp0);
    fq2_add(&tmp6, &r->x, &tmp4);
    fq2_sqr(&tmp5, &tmp4);
    fq2_sqr(&zsq, &r->z);
    fq2_sub(&r->x, &tmp5, &tmp3); fq2_sub(&r->x, &r->x, &tmp3);
    fq2_add(&r->z, &r->z, &r->y); fq2_sqr(&r->z, &r->z); fq2_sub(&r->z, &r->z, &tmp1); fq2_sub(&r->z, &r->z, &zsq);
    fq2_sub(&r->y, &tmp3, &r->x); fq2_mul(&r->y, &r->y, &tmp4);
    fq2_dbl(&tmp2, &tmp2); fq2_dbl(&tmp2, &tmp2); fq2_dbl(&tmp2, &tmp2);
    fq2_sub(&r->y, &r->y, &tmp2);
    fq2_mul(&tmp3, &tmp4, &zsq); fq2_dbl(&tmp3, &tmp3); fq2_neg(&tmp3, &tmp3);
    fq2_sqr(&tmp6, &tmp6); fq2_sub(&tmp6, &tmp6, &tmp0); fq2_sub(&tmp6, &tmp6, &tmp5);
    fq2_dbl(&tmp1, &tmp1); fq2_dbl(&tmp1, &tmp1);
    fq2_sub(&tmp6, &tmp6, &tmp1);
    fq2_mul(&tmp0, &r->z, &zsq); fq2_dbl(&tmp0, &tmp0);
    out[0] = tmp0; out[1] = tmp3; out[2] = tmp6;
}
static void addition_step(g2_jac *r, const g2_aff *q, fq2 out[3]) {                                                                /* g2.go:710-772 */
    fq2 zsq, ysq, t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
    fq2_sqr(&zsq, &r->z);
    fq2_sqr(&ysq, &q->y);
    fq2_mul(&t0, &zsq, &q->x);
    fq2_add(&t1, &q->y, &r->z); fq2_sqr(&t1, &t1); fq2_sub(&t1, &t1, &ysq); fq2_sub(&t1, &t1, &zsq); fq2_mul(&t1, &t1, &zsq);
    fq2_sub(&t2, &t0, &r->x);
    fq2_sqr(&t3, &t2);
    fq2_dbl(&t4, &t3); fq2_dbl(&t4, &t4);
    fq2_mul(&t5, &t4, &t2);
    fq2_sub(&t6, &t1, &r->y); fq2_sub(&t6, &t6, &r->y);
    fq2_mul(&t9, &t6, &q->x);
    fq2_mul(&t7, &t4, &r->x);
    fq2_sqr(&r->x, &t6); fq2_sub(&r->x, &r->x, &t5); fq2_sub(&r->x, &r->x, &t7); fq2_sub(&r->x, &r->x, &t7);
    fq2_add(&r->z, &r->z, &t2); fq2_sqr(&r->z, &r->z); fq2_sub(&r->z, &r->z, &zsq); fq2_sub(&r->z, &r->z, &t3);
    fq2_add(&t10, &q->y, &r->z);
    fq2_sub(&t8, &t7, &r->x); fq2_mul(&t8, &t8, &t6);
    fq2_mul(&t0, &r->y, &t5); fq2_dbl(&t0, &t0);
    fq2_sub(&r->y, &t8, &t0);
    fq2_sqr(&t10, &t10); fq2_sub(&t10, &t10, &ysq);
    fq2_sqr(&zsq, &r->z);
    fq2_sub(&t10, &t10, &zsq);
    fq2_dbl(&t9, &t9); fq2_sub(&t9, &t9, &t10);
    fq2_dbl(&t10, &r->z);
    fq2_neg(&t6, &t6); fq2_dbl(&t6, &t6);
    out[0] = t10; out[1] = t6; out[2] = t9;
}
static void g2_prepare(g2_prepared *p, const g2_aff *q) {                                                                          /* g2.go:650-801 */
    if (q->inf) { p->inf = 1; return; }
    p->inf = 0;
    g2_jac r; r.x = q->x; r.y = q->y; r.z = fq2_one();
    const u64 xr = BLS_X >> 1;
    int n = 0;
    for (int i = 61; i >= 0; i--) {          /* bits below the leading one of |x|>>1 (bit 62) */
        doubling_step(&r, p->c[n++]);
        if ((xr >> i) & 1) addition_step(&r, q, p->c[n++]);
    }
    doubling_step(&r, p->c[n++]);
}

/* ------------------------------------------------------------------------------------------
 * L5: pairing (pairing.go)
 * ---------------------------------------------------------------------------------------- */
static void ell(fq12 *f, const fq2 coeffs[3], const g1_aff *p) {                                                                   /* pairing.go:28-39 */
    fq2 c0, c1;
    fq2_mul_fq(&c0, &coeffs[0], &p->y);
    fq2_mul_fq(&c1, &coeffs[1], &p->x);
    fq12_mul_by_014(f, f, &coeffs[2], &c1, &c0);
}
/* pairing.go:16-75.  Items at infinity are skipped (the Go code leaves nil entries and would panic
 * when indexing them; here a skipped pair contributes 1, the mathematically defined value). */
static void miller_loop(fq12 *out, const g1_aff *ps, const g2_prepared *qs, int n) {
    fq12 f = fq12_one();
    const u64 xr = BLS_X >> 1;
    int idx = 0;
    for (int i = 61; i >= 0; i--) {
        for (int k = 0; k < n; k++) if (!ps[k].inf && !qs[k].inf) ell(&f, qs[k].c[idx], &ps[k]);
        idx++;
        if ((xr >> i) & 1) {
            for (int k = 0; k < n; k++) if (!ps[k].inf && !qs[k].inf) ell(&f, qs[k].c[idx], &ps[k]);
            idx++;
        }
        fq12_sqr(&f, &f);
    }
    for (int k = 0; k < n; k++) if (!ps[k].inf && !qs[k].inf) ell(&f, qs[k].c[idx], &ps[k]);
    fq12_conj(out, &f);
}
static void exp_by_x(fq12 *r, const fq12 *f, u64 x) { fq12_exp_u64(r, f, x); fq12_conj(r, r); }                                   /* pairing.go:92-98 */
static int final_exponentiation(fq12 *out, const fq12 *in) {                                                                       /* pairing.go:79-129 */
    fq12 f1, f2, r, y0, y1, y2, y3;
    fq12_conj(&f1, in);
    if (!fq12_inverse(&f2, in)) return 0;
    fq12_mul(&r, &f1, &f2);
    f2 = r;
    fq12_frob(&r, &r, 2);
    fq12_mul(&r, &r, &f2);
    u64 x = BLS_X;
    fq12_sqr(&y0, &r);
    exp_by_x(&y1, &y0, x);
    exp_by_x(&y2, &y1, x >> 1);
    fq12_conj(&y3, &r);
    fq12_mul(&y1, &y1, &y3);
    fq12_conj(&y1, &y1);
    fq12_mul(&y1, &y1, &y2);
    exp_by_x(&y2, &y1, x);
    exp_by_x(&y3, &y2, x);
    fq12_conj(&y1, &y1);
    fq12_mul(&y3, &y3, &y1);
    fq12_conj(&y1, &y1);
    fq12_frob(&y1, &y1, 3);
    fq12_frob(&y2, &y2, 2);
    fq12_mul(&y1, &y1, &y2);
    exp_by_x(&y2, &y3, x);
    fq12_mul(&y2, &y2, &y0);
    fq12_mul(&y2, &y2, &r);
    fq12_mul(&y1, &y1, &y2);
    fq12_frob(&y3, &y3, 1);
    fq12_mul(out, &y1, &y3);
    return 1;
}
static int pairing_aff(fq12 *out, const g1_aff *p, const g2_aff *q) {                                                              /* pairing.go:132-136 */
    g2_prepared *prep = (g2_prepared *)malloc(sizeof(g2_prepared));
    g2_prepare(prep, q);
    fq12 f;
    miller_loop(&f, p, prep, 1);
    free(prep);
    return final_exponentiation(out, &f);
}
static int compare_two_pairings(const g1_aff *p1, const g2_aff *q1, const g1_aff *p2, const g2_aff *q2) {                          /* pairing.go:140-147 */
    g2_prepared *prep = (g2_prepared *)malloc(2 * sizeof(g2_prepared));
    g1_aff ps[2];
    ps[0] = *p1; ps[1] = *p2;
    if (!ps[1].inf) fq_neg(&ps[1].y, &ps[1].y);
    g2_prepare(&prep[0], q1);
    g2_prepare(&prep[1], q2);
    fq12 f, e, one = fq12_one();
    miller_loop(&f, ps, prep, 2);
    free(prep);
    if (!final_exponentiation(&e, &f)) return 0;
    return fq12_eq(&e, &one);
}

/* ------------------------------------------------------------------------------------------
 * SHA-256 (FIPS 180-4) -- the reference uses Go's crypto/sha256 (hash.go:4)
 * ---------------------------------------------------------------------------------------- */
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t h[8], const u8 *p) {
    uint32_t w[64], a, b, c, d, e, f, g, hh;
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    a = h[0]; b = h[1]; c = h[2]; d = h[3]; e = h[4]; f = h[5]; g = h[6]; hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + SHA_K[i] + w[i];
        uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22), mj = (a & b) ^ (a & c) ^ (b & c), t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
/* digest of the concatenation of up to 3 byte strings */
static void sha256_3(u8 out[32], const u8 *a, size_t na, const u8 *b, size_t nb, const u8 *c, size_t nc) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    u8 buf[64];
    size_t fill = 0, total = na + nb + nc;
    const u8 *parts[3] = {a, b, c};
    size_t lens[3] = {na, nb, nc};
    for (int k = 0; k < 3; k++)
        for (size_t i = 0; i < lens[k]; i++) { buf[fill++] = parts[k][i]; if (fill == 64) { sha256_block(h, buf); fill = 0; } }
    buf[fill++] = 0x80;
    if (fill > 56) { while (fill < 64) buf[fill++] = 0; sha256_block(h, buf); fill = 0; }
    while (fill < 56) buf[fill++] = 0;
    u64 bits = (u64)total * 8;
    for (int i = 0; i < 8; i++) buf[56 + i] = (u8)(bits >> (56 - 8 * i));
    sha256_block(h, buf);
    for (int i = 0; i < 8; i++) { out[4 * i] = (u8)(h[i] >> 24); out[4 * i + 1] = (u8)(h[i] >> 16); out[4 * i + 2] = (u8)(h[i] >> 8); out[4 * i + 3] = (u8)h[i]; }
}
API void rc_sha256(const u8 *msg, size_t n, u8 out[32]) { sha256_3(out, msg, n, 0, 0, 0, 0); }

/* ------------------------------------------------------------------------------------------
 * L6: hash to curve (hash.go, g1.go:614-714, g2.go:883-1085)
 * ---------------------------------------------------------------------------------------- */
/* 64-byte big-endian integer mod q, as a Montgomery Fq (hash.go:66-71: big.Int Mod then FQReprToFQ) */
static void fq_from_wide_be(fq *r, const u8 t[64]) {
    u64 hi[6] = {0}, lo[6] = {0};
    repr_from_be(hi, t, 4);           /* top 256 bits  */
    repr_from_be(lo, t + 32, 4);      /* low 256 bits  */
    fq fh, fl;
    fq_from_repr(&fh, hi);
    fq_from_repr(&fl, lo);
    fq_mul(&fh, &fh, &RC_TWO256);
    fq_add(r, &fh, &fl);
}
/* hash.go:41-72 (hp) with cipher-suite byte already part of `msg`; prefix lets callers avoid a copy */
static void hash_msg_prime(u8 mp[33], const u8 *prefix, size_t npre, const u8 *msg, size_t n, u8 ctr) {
    sha256_3(mp, prefix, npre, msg, n, 0, 0);
    mp[32] = ctr;
}
static void hp(fq *r, const u8 *prefix, size_t npre, const u8 *msg, size_t n, u8 ctr) {
    u8 mp[33], t[64], tail[2];
    hash_msg_prime(mp, prefix, npre, msg, n, ctr);
    for (int j = 1; j <= 2; j++) { tail[0] = 0x01; tail[1] = (u8)j; sha256_3(t + 32 * (j - 1), mp, 33, tail, 2, 0, 0); }
    fq_from_wide_be(r, t);
}
static void hp2(fq2 *r, const u8 *prefix, size_t npre, const u8 *msg, size_t n, u8 ctr) {                                          /* hash.go:74-113 */
    u8 mp[33], t[64], tail[2];
    hash_msg_prime(mp, prefix, npre, msg, n, ctr);
    fq *dst[2] = {&r->c0, &r->c1};
    for (int i = 1; i <= 2; i++) {
        for (int j = 1; j <= 2; j++) { tail[0] = (u8)i; tail[1] = (u8)j; sha256_3(t + 32 * (j - 1), mp, 33, tail, 2, 0, 0); }
        fq_from_wide_be(dst[i - 1], t);
    }
}
static void sign_fq(fq *r, const fq *f) {                                                                                          /* g1.go:621-626 */
    u64 x[6];
    fq_to_repr(x, f);
    *r = (repr_cmp(x, RC_QM1O2, 6) > 0) ? RC_NEGONE : RC_ONE;
}
static void swu_g1_helper(g1_aff *out, const fq *t) {                                                                              /* g1.go:628-714 */
    fq ndc, tsq, t2, n1t, x0, inv;
    fq_sqr(&ndc, &RC_NEGONE);
    fq_sqr(&tsq, t);
    fq_sqr(&t2, &tsq);
    fq_mul(&ndc, &ndc, &t2);
    fq_mul(&n1t, &RC_NEGONE, &tsq);
    fq_add(&ndc, &ndc, &n1t);
    if (fq_is_zero(&ndc)) {
        fq xia; fq_mul(&xia, &RC_NEGONE, &RC_ELLPA);
        fq_inverse(&inv, &xia);
        fq_mul(&x0, &RC_ELLPB, &inv);
    } else {
        fq apc, nb;
        fq_mul(&apc, &RC_ELLPA, &ndc);
        fq_add(&ndc, &ndc, &RC_ONE);
        fq_neg(&nb, &RC_ELLPB);
        fq_mul(&x0, &nb, &ndc);
        fq_inverse(&inv, &apc);
        fq_mul(&x0, &x0, &inv);
    }
    fq gx0, ax, x, y;
    fq_sqr(&gx0, &x0); fq_mul(&gx0, &gx0, &x0);
    fq_mul(&ax, &RC_ELLPA, &x0);
    fq_add(&gx0, &gx0, &ax); fq_add(&gx0, &gx0, &RC_ELLPB);
    if (fq_sqrt(&y, &gx0)) {
        x = x0;
    } else {
        fq x1, gx1;
        fq_mul(&x1, &RC_NEGONE, &tsq); fq_mul(&x1, &x1, &x0);
        fq_mul(&ax, &RC_ELLPA, &x1);
        fq_sqr(&gx1, &x1); fq_mul(&gx1, &gx1, &x1); fq_add(&gx1, &gx1, &ax); fq_add(&gx1, &gx1, &RC_ELLPB);
        fq_sqrt(&y, &gx1);            /* "this should never happen" otherwise (g1.go:698-700) */
        x = x1;
    }
    fq st, sy;
    sign_fq(&st, t); sign_fq(&sy, &y);
    fq_mul(&sy, &sy, &st);
    fq_mul(&y, &y, &sy);
    out->x = x; out->y = y; out->inf = 0;
}
static void horner_fq(fq *r, const fq *coeffs, int n, const fq *x) {
    fq v = coeffs[n - 1];
    for (int i = n - 2; i >= 0; i--) { fq_mul(&v, &v, x); fq_add(&v, &v, &coeffs[i]); }
    *r = v;
}
static void iso11(g1_aff *out, const g1_aff *p) {                                                                                  /* hash.go:185-206 */
    fq xn, xd, yn, yd, inv;
    horner_fq(&xn, RC_XNUM11, 12, &p->x); horner_fq(&xd, RC_XDEN11, 11, &p->x);
    horner_fq(&yn, RC_YNUM11, 16, &p->x); horner_fq(&yd, RC_YDEN11, 16, &p->x);
    fq_inverse(&inv, &xd); fq_mul(&out->x, &xn, &inv);
    fq_mul(&yn, &p->y, &yn);
    fq_inverse(&inv, &yd); fq_mul(&out->y, &yn, &inv);
    out->inf = 0;
}
static void clear_h(g1_aff *out, const g1_aff *p) {                                                                                /* hash.go:306-309 */
    u64 x[1] = {BLS_X};
    g1_jac xp, s;
    g1_aff_mul(&xp, p, x, 1);
    g1_add_affine(&s, &xp, p);
    g1_to_affine(out, &s);
}
static void swu_map_g1(g1_aff *out, const fq *t1, const fq *t2) {                                                                  /* hash.go:311-321 */
    g1_aff pp, pp2, iso;
    swu_g1_helper(&pp, t1);
    if (t2) {
        swu_g1_helper(&pp2, t2);
        g1_jac j, s;
        g1_to_jac(&j, &pp);
        g1_add_affine(&s, &j, &pp2);
        g1_to_affine(&pp, &s);
    }
    iso11(&iso, &pp);
    clear_h(out, &iso);
}
static void hash_g1(g1_aff *out, const u8 *msg, size_t n) {                                                                        /* hash.go:326-331 */
    const u8 cs = 0x01;
    fq t1, t2;
    hp(&t1, &cs, 1, msg, n, 0);
    hp(&t2, &cs, 1, msg, n, 1);
    swu_map_g1(out, &t1, &t2);
}
static int sign_fq2(const fq2 *f) {                                                                                                /* g2.go:916-931 */
    u64 c1[6], c0[6];
    fq_to_repr(c1, &f->c1); fq_to_repr(c0, &f->c0);
    if (repr_cmp(c1, RC_QM1O2, 6) > 0) return -1;
    if (!repr_is_zero(c1, 6)) return 1;
    if (repr_cmp(c0, RC_QM1O2, 6) > 0) return -1;
    return 1;
}
static int swu_g2_helper(g2_aff *out, const fq2 *t) {                                                                              /* g2.go:933-1031 */
    fq2 ndc, tsq, t4, n1t, x0, inv, one = fq2_one();
    fq2_sqr(&ndc, &RC_NQR);
    fq2_sqr(&tsq, t);
    fq2_sqr(&t4, &tsq);
    fq2_mul(&ndc, &ndc, &t4);
    fq2_mul(&n1t, &RC_NQR, &tsq);
    fq2_add(&ndc, &ndc, &n1t);
    if (fq2_is_zero(&ndc)) {
        fq2 xia; fq2_mul(&xia, &RC_NQR, &RC_ELL2PA);
        fq2_inverse(&inv, &xia);
        fq2_mul(&x0, &RC_ELL2PB, &inv);
    } else {
        fq2 apc, nb;
        fq2_mul(&apc, &RC_ELL2PA, &ndc);
        fq2_neg(&nb, &RC_ELL2PB);
        fq2_add(&ndc, &ndc, &one);
        fq2_mul(&x0, &nb, &ndc);
        fq2_inverse(&inv, &apc);
        fq2_mul(&x0, &x0, &inv);
    }
    fq2 gx0, ax, s, chk;
    fq2_sqr(&gx0, &x0); fq2_mul(&gx0, &gx0, &x0);
    fq2_mul(&ax, &RC_ELL2PA, &x0);
    fq2_add(&gx0, &gx0, &ax); fq2_add(&gx0, &gx0, &RC_ELL2PB);
    if (fq2_sqrt(&s, &gx0)) {
        fq2_sqr(&chk, &s);
        if (fq2_eq(&chk, &gx0)) {
            if (sign_fq2(t) != sign_fq2(&s)) fq2_neg(&s, &s);
            out->x = x0; out->y = s; out->inf = 0;
            return 1;
        }
    }
    fq2 tcu, t6, x1, gx1, y1;
    fq2_mul(&tcu, &tsq, t);
    fq2_sqr(&t6, &tcu);
    fq2_mul(&x1, &RC_NQR, &tsq); fq2_mul(&x1, &x1, &x0);
    fq2_sqr(&gx1, &RC_NQR); fq2_mul(&gx1, &gx1, &RC_NQR); fq2_mul(&gx1, &gx1, &t6); fq2_mul(&gx1, &gx1, &gx0);
    if (!fq2_sqrt(&y1, &gx1)) return 0;     /* "This should never happen!" (g2.go:1013-1015) */
    fq2_sqr(&chk, &y1);
    if (fq2_eq(&chk, &gx1)) {
        if (sign_fq2(t) != sign_fq2(&y1)) fq2_neg(&y1, &y1);
        out->x = x1; out->y = y1; out->inf = 0;
        return 1;
    }
    return 0;
}
static void horner_fq2(fq2 *r, const fq2 *coeffs, int n, const fq2 *x) {
    fq2 v = coeffs[n - 1];
    for (int i = n - 2; i >= 0; i--) { fq2_mul(&v, &v, x); fq2_add(&v, &v, &coeffs[i]); }
    *r = v;
}
static void iso3(g2_aff *out, const g2_aff *p) {                                                                                   /* hash.go:282-303 */
    fq2 xn, xd, yn, yd, inv;
    horner_fq2(&xn, RC_XNUM3, 4, &p->x); horner_fq2(&xd, RC_XDEN3, 3, &p->x);
    horner_fq2(&yn, RC_YNUM3, 4, &p->x); horner_fq2(&yd, RC_YDEN3, 4, &p->x);
    fq2_inverse(&inv, &xd); fq2_mul(&out->x, &xn, &inv);
    fq2_mul(&yn, &p->y, &yn);
    fq2_inverse(&inv, &yd); fq2_mul(&out->y, &yn, &inv);
    out->inf = 0;
}
static void psi(g2_aff *out, const g2_aff *g) {                                                                                    /* hash.go:341-366 */
    fq2 qix, qiy, ny;
    fq2_mul(&qix, &RC_IWSC, &g->x);
    fq_mul(&qix.c0, &qix.c0, &RC_KQIX);
    fq_mul(&qix.c1, &qix.c1, &RC_KQIX);
    fq_neg(&qix.c1, &qix.c1);
    fq2_mul(&out->x, &RC_NQR, &qix);
    fq2_mul(&qiy, &RC_IWSC, &g->y);
    fq s, d;
    fq_add(&s, &qiy.c0, &qiy.c1); fq_mul(&s, &s, &RC_KQIY);
    fq_sub(&d, &qiy.c0, &qiy.c1); fq_mul(&d, &d, &RC_KQIY);
    qiy.c0 = s; qiy.c1 = d;
    fq2_mul(&ny, &RC_NQR, &qiy);
    out->y = ny; out->inf = 0;
}
static void clear_h2(g2_aff *out, const g2_aff *p) {                                                                               /* hash.go:368-389 */
    u64 x[1] = {BLS_X};
    g2_jac work, tmp;
    g2_aff mpsi, negp, p2, pp;
    g2_aff_mul(&work, p, x, 1);
    g2_add_affine(&tmp, &work, p); work = tmp;
    psi(&mpsi, p); fq2_neg(&mpsi.y, &mpsi.y);
    g2_add_affine(&tmp, &work, &mpsi); work = tmp;
    g2_jac_mul(&tmp, &work, x, 1); work = tmp;
    g2_add_affine(&tmp, &work, &mpsi); work = tmp;
    negp = *p; fq2_neg(&negp.y, &negp.y);
    g2_add_affine(&tmp, &work, &negp); work = tmp;
    g2_jac pj, dj;
    g2_to_jac(&pj, p);
    g2_double(&dj, &pj);
    g2_to_affine(&p2, &dj);
    psi(&pp, &p2); psi(&pp, &pp);
    g2_add_affine(&tmp, &work, &pp); work = tmp;
    g2_to_affine(out, &work);
}
static int swu_map_g2(g2_aff *out, const fq2 *t1, const fq2 *t2) {                                                                 /* hash.go:391-402 */
    g2_aff pp, pp2, iso;
    if (!swu_g2_helper(&pp, t1)) return 0;
    if (t2) {
        if (!swu_g2_helper(&pp2, t2)) return 0;
        g2_jac j, s;
        g2_to_jac(&j, &pp);
        g2_add_affine(&s, &j, &pp2);
        g2_to_affine(&pp, &s);
    }
    iso3(&iso, &pp);
    clear_h2(out, &iso);
    return 1;
}
static int hash_g2(g2_aff *out, const u8 *msg, size_t n) {                                                                         /* hash.go:405-411 */
    const u8 cs = 0x01;
    fq2 t1, t2;
    hp2(&t1, &cs, 1, msg, n, 0);
    hp2(&t2, &cs, 1, msg, n, 1);
    return swu_map_g2(out, &t1, &t2);
}
static void hash_g2_with_domain(g2_jac *out, const u8 msg[32], const u8 domain[8]) {                                               /* g2.go:1041-1085 */
    u8 d[32], tag;
    u64 re[6] = {0}, im[6] = {0};
    tag = 0x01; sha256_3(d, msg, 32, domain, 8, &tag, 1); repr_from_be(re, d, 4);
    tag = 0x02; sha256_3(d, msg, 32, domain, 8, &tag, 1); repr_from_be(im, d, 4);
    fq2 x0, gx0, y0, one = fq2_one();
    fq_from_repr(&x0.c0, re);
    fq_from_repr(&x0.c1, im);
    for (;;) {
        fq2_sqr(&gx0, &x0); fq2_mul(&gx0, &gx0, &x0); fq2_add(&gx0, &gx0, &RC_B2);
        if (fq2_sqrt(&y0, &gx0)) {
            if (!fq2_parity(&y0)) fq2_neg(&y0, &y0);
            g2_aff a; a.x = x0; a.y = y0; a.inf = 0;
            g2_aff_mul(out, &a, RC_G2COF, 8);
            return;
        }
        fq2_add(&x0, &x0, &one);
    }
}

/* ------------------------------------------------------------------------------------------
 * Wire formats (g1.go:157-249, g2.go:172-295)
 * ---------------------------------------------------------------------------------------- */
static void fq_to_be(u8 out[48], const fq *a) { u64 r[6]; fq_to_repr(r, a); repr_to_be(out, r, 6); }
static void fq_from_be(fq *r, const u8 in[48]) { u64 x[6]; repr_from_be(x, in, 6); fq_from_repr(r, x); }
static void g1_read_affine(g1_aff *p, const u8 in[96]) { fq_from_be(&p->x, in); fq_from_be(&p->y, in + 48); p->inf = 0; }
static void g2_read_affine(g2_aff *p, const u8 in[192]) {
    fq_from_be(&p->x.c0, in); fq_from_be(&p->x.c1, in + 48); fq_from_be(&p->y.c0, in + 96); fq_from_be(&p->y.c1, in + 144); p->inf = 0;
}
static void g1_write_affine(u8 out[96], const g1_aff *p) { fq_to_be(out, &p->x); fq_to_be(out + 48, &p->y); }                     /* g1.go:157-167 */
static void g2_write_affine(u8 out[192], const g2_aff *p) {                                                                        /* g2.go:172-186 */
    fq_to_be(out, &p->x.c0); fq_to_be(out + 48, &p->x.c1); fq_to_be(out + 96, &p->y.c0); fq_to_be(out + 144, &p->y.c1);
}
static int g1_from_x(g1_aff *out, const fq *x, int greatest) {                                                                     /* g1.go:111-132 */
    fq x3b, y, negy;
    fq_sqr(&x3b, x); fq_mul(&x3b, &x3b, x); fq_add(&x3b, &x3b, &RC_B);
    if (!fq_sqrt(&y, &x3b)) return 0;
    fq_neg(&negy, &y);
    out->x = *x; out->y = ((fq_cmp(&y, &negy) < 0) != greatest) ? y : negy; out->inf = 0;
    return 1;
}
static int g2_from_x(g2_aff *out, const fq2 *x, int greatest) {                                                                    /* g2.go:149-169 */
    fq2 x3b, y, negy;
    fq2_sqr(&x3b, x); fq2_mul(&x3b, &x3b, x); fq2_add(&x3b, &x3b, &RC_B2);
    if (!fq2_sqrt(&y, &x3b)) return 0;
    fq2_neg(&negy, &y);
    out->x = *x; out->y = ((fq2_cmp(&y, &negy) < 0) != greatest) ? y : negy; out->inf = 0;
    return 1;
}
/* error codes: 0 ok, 1 unexpected compression mode, 2 bad infinity encoding, 3 not on curve, 4 not in subgroup */
static int g1_decompress_unchecked(g1_aff *out, const u8 in[48]) {                                                                 /* g1.go:201-227 */
    u8 c[48]; memcpy(c, in, 48);
    if (!(c[0] & 0x80)) return 1;
    if (c[0] & 0x40) {
        c[0] &= 0x3f;
        for (int i = 0; i < 48; i++) if (c[i]) return 2;
        out->x = FQ_ZERO; out->y = RC_ONE; out->inf = 1;
        return 0;
    }
    int greatest = (c[0] & 0x20) != 0;
    c[0] &= 0x1f;
    fq x; fq_from_be(&x, c);
    return g1_from_x(out, &x, greatest) ? 0 : 3;
}
static int g2_decompress_unchecked(g2_aff *out, const u8 in[96]) {                                                                 /* g2.go:234-267 */
    u8 c[96]; memcpy(c, in, 96);
    if (!(c[0] & 0x80)) return 1;
    if (c[0] & 0x40) {
        c[0] &= 0x3f;
        for (int i = 0; i < 96; i++) if (c[i]) return 2;
        out->x = FQ2_ZERO; out->y = fq2_one(); out->inf = 1;
        return 0;
    }
    int greatest = (c[0] & 0x20) != 0;
    c[0] &= 0x1f;
    fq2 x; fq_from_be(&x.c1, c); fq_from_be(&x.c0, c + 48);
    return g2_from_x(out, &x, greatest) ? 0 : 3;
}
static void g1_compress(u8 out[48], const g1_aff *a) {                                                                             /* g1.go:230-249 */
    memset(out, 0, 48);
    if (a->inf) out[0] |= 0x40;
    else { fq_to_be(out, &a->x); if (fq_parity(&a->y)) out[0] |= 0x20; }
    out[0] |= 0x80;
}
static void g2_compress(u8 out[96], const g2_aff *a) {                                                                             /* g2.go:269-289 */
    memset(out, 0, 96);
    if (a->inf) out[0] |= 0x40;
    else { fq_to_be(out, &a->x.c1); fq_to_be(out + 48, &a->x.c0); if (fq2_parity(&a->y)) out[0] |= 0x20; }
    out[0] |= 0x80;
}
static int g1_in_subgroup(const g1_aff *a) { g1_jac t; g1_aff_mul(&t, a, RC_RORDER, 4); g1_aff_mul(&t, a, RC_RORDER, 4); return g1_jac_is_zero(&t); }   /* g1.go:137-141 (computed twice there) */
static int g2_in_subgroup(const g2_aff *a) { g2_jac t; g2_aff_mul(&t, a, RC_RORDER, 4); return g2_jac_is_zero(&t); }                                 /* g2.go:293-295 */

/* ==========================================================================================
 * Exported test surface.  Fq values cross this boundary either as 6 u64 Montgomery limbs
 * (little-endian limb order -- the reference's in-memory FQ) or as 48-byte big-endian normal form.
 * ======================================================================================== */
API void rc_fq_from_repr(const u64 repr[6], u64 out[6]) { fq r; fq_from_repr(&r, repr); memcpy(out, r.l, 48); }
API void rc_fq_to_repr(const u64 in[6], u64 out[6]) { fq a; memcpy(a.l, in, 48); fq_to_repr(out, &a); }
#define FQ_BINOP(name) API void rc_fq_##name(const u64 a[6], const u64 b[6], u64 out[6]) { fq x, y, r; memcpy(x.l, a, 48); memcpy(y.l, b, 48); fq_##name(&r, &x, &y); memcpy(out, r.l, 48); }
FQ_BINOP(add) FQ_BINOP(sub) FQ_BINOP(mul)
#define FQ_UNOP(name) API void rc_fq_##name(const u64 a[6], u64 out[6]) { fq x, r; memcpy(x.l, a, 48); fq_##name(&r, &x); memcpy(out, r.l, 48); }
FQ_UNOP(sqr) FQ_UNOP(neg) FQ_UNOP(dbl)
API int rc_fq_inverse(const u64 a[6], u64 out[6]) { fq x, r = FQ_ZERO; memcpy(x.l, a, 48); int ok = fq_inverse(&r, &x); memcpy(out, r.l, 48); return ok; }
API int rc_fq_sqrt(const u64 a[6], u64 out[6]) { fq x, r = FQ_ZERO; memcpy(x.l, a, 48); int ok = fq_sqrt(&r, &x); memcpy(out, r.l, 48); return ok; }
#define FQ2_BINOP(name) API void rc_fq2_##name(const u64 a[12], const u64 b[12], u64 out[12]) { fq2 x, y, r; memcpy(&x, a, 96); memcpy(&y, b, 96); fq2_##name(&r, &x, &y); memcpy(out, &r, 96); }
FQ2_BINOP(add) FQ2_BINOP(sub) FQ2_BINOP(mul)
#define FQ2_UNOP(name) API void rc_fq2_##name(const u64 a[12], u64 out[12]) { fq2 x, r; memcpy(&x, a, 96); fq2_##name(&r, &x); memcpy(out, &r, 96); }
FQ2_UNOP(sqr) FQ2_UNOP(neg) FQ2_UNOP(dbl) FQ2_UNOP(mul_nr)
API int rc_fq2_inverse(const u64 a[12], u64 out[12]) { fq2 x, r = FQ2_ZERO; memcpy(&x, a, 96); int ok = fq2_inverse(&r, &x); memcpy(out, &r, 96); return ok; }
API int rc_fq2_sqrt(const u64 a[12], u64 out[12]) { fq2 x, r = FQ2_ZERO; memcpy(&x, a, 96); int ok = fq2_sqrt(&r, &x); memcpy(out, &r, 96); return ok; }
API void rc_fq2_frobenius(const u64 a[12], unsigned power, u64 out[12]) { fq2 x, r; memcpy(&x, a, 96); fq2_frob(&r, &x, power); memcpy(out, &r, 96); }
API void rc_fq6_mul(const u64 a[36], const u64 b[36], u64 out[36]) { fq6 x, y, r; memcpy(&x, a, 288); memcpy(&y, b, 288); fq6_mul(&r, &x, &y); memcpy(out, &r, 288); }
API void rc_fq6_sqr(const u64 a[36], u64 out[36]) { fq6 x, r; memcpy(&x, a, 288); fq6_sqr(&r, &x); memcpy(out, &r, 288); }
API int rc_fq6_inverse(const u64 a[36], u64 out[36]) { fq6 x, r; memcpy(&x, a, 288); memset(&r, 0, 288); int ok = fq6_inverse(&r, &x); memcpy(out, &r, 288); return ok; }
API void rc_fq6_frobenius(const u64 a[36], unsigned power, u64 out[36]) { fq6 x, r; memcpy(&x, a, 288); fq6_frob(&r, &x, power); memcpy(out, &r, 288); }
API void rc_fq12_mul(const u64 a[72], const u64 b[72], u64 out[72]) { fq12 x, y, r; memcpy(&x, a, 576); memcpy(&y, b, 576); fq12_mul(&r, &x, &y); memcpy(out, &r, 576); }
API void rc_fq12_sqr(const u64 a[72], u64 out[72]) { fq12 x, r; memcpy(&x, a, 576); fq12_sqr(&r, &x); memcpy(out, &r, 576); }
API int rc_fq12_inverse(const u64 a[72], u64 out[72]) { fq12 x, r; memcpy(&x, a, 576); memset(&r, 0, 576); int ok = fq12_inverse(&r, &x); memcpy(out, &r, 576); return ok; }
API void rc_fq12_frobenius(const u64 a[72], unsigned power, u64 out[72]) { fq12 x, r; memcpy(&x, a, 576); fq12_frob(&r, &x, power); memcpy(out, &r, 576); }
API void rc_fq12_mul_by_014(const u64 a[72], const u64 c0[12], const u64 c1[12], const u64 c4[12], u64 out[72]) {
    fq12 x, r; fq2 k0, k1, k4; memcpy(&x, a, 576); memcpy(&k0, c0, 96); memcpy(&k1, c1, 96); memcpy(&k4, c4, 96);
    fq12_mul_by_014(&r, &x, &k0, &k1, &k4); memcpy(out, &r, 576);
}
/* sparse Fq6 products (fq6.go:40-57, 60-90) and the normal-form comparisons (fq.go:134-137, 269-273; fq2.go:31-37, 256-260) */
API void rc_fq6_mul_by_1(const u64 a[36], const u64 c1[12], u64 out[36]) { fq6 x, r; fq2 k1; memcpy(&x, a, 288); memcpy(&k1, c1, 96); fq6_mul_by_1(&r, &x, &k1); memcpy(out, &r, 288); }
API void rc_fq6_mul_by_01(const u64 a[36], const u64 c0[12], const u64 c1[12], u64 out[36]) { fq6 x, r; fq2 k0, k1; memcpy(&x, a, 288); memcpy(&k0, c0, 96); memcpy(&k1, c1, 96); fq6_mul_by_01(&r, &x, &k0, &k1); memcpy(out, &r, 288); }
API int rc_fq_cmp(const u64 a[6], const u64 b[6]) { fq x, y; memcpy(x.l, a, 48); memcpy(y.l, b, 48); return fq_cmp(&x, &y); }
API int rc_fq_parity(const u64 a[6]) { fq x; memcpy(x.l, a, 48); return fq_parity(&x); }
API int rc_fq2_cmp(const u64 a[12], const u64 b[12]) { fq2 x, y; memcpy(&x, a, 96); memcpy(&y, b, 96); return fq2_cmp(&x, &y); }
API int rc_fq2_parity(const u64 a[12]) { fq2 x; memcpy(&x, a, 96); return fq2_parity(&x); }
API void rc_fq12_exp_u64(const u64 a[72], u64 e, u64 out[72]) { fq12 x, r; memcpy(&x, a, 576); fq12_exp_u64(&r, &x, e); memcpy(out, &r, 576); }
API int rc_final_exponentiation(const u64 a[72], u64 out[72]) { fq12 x, r; memcpy(&x, a, 576); memset(&r, 0, 576); int ok = final_exponentiation(&r, &x); memcpy(out, &r, 576); return ok; }

/* Jacobian points cross as 3 (G1: 18 u64) / 3x2 (G2: 36 u64) Montgomery coordinates */
API void rc_g1_double(const u64 p[18], u64 out[18]) { g1_jac a, r; memcpy(&a, p, 144); g1_double(&r, &a); memcpy(out, &r, 144); }
API void rc_g1_add(const u64 p[18], const u64 q[18], u64 out[18]) { g1_jac a, b, r; memcpy(&a, p, 144); memcpy(&b, q, 144); g1_add(&r, &a, &b); memcpy(out, &r, 144); }
API void rc_g2_double(const u64 p[36], u64 out[36]) { g2_jac a, r; memcpy(&a, p, 288); g2_double(&r, &a); memcpy(out, &r, 288); }
API void rc_g2_add(const u64 p[36], const u64 q[36], u64 out[36]) { g2_jac a, b, r; memcpy(&a, p, 288); memcpy(&b, q, 288); g2_add(&r, &a, &b); memcpy(out, &r, 288); }
/* Jacobian -> affine big-endian bytes; returns 1 if infinity (bytes zeroed) */
API int rc_g1_jac_to_affine_bytes(const u64 p[18], u8 out[96]) { g1_jac a; g1_aff r; memcpy(&a, p, 144); g1_to_affine(&r, &a); if (r.inf) { memset(out, 0, 96); return 1; } g1_write_affine(out, &r); return 0; }
/* n points in one call (bench.py `marshal`: what the affine entry points cost a Go shim per point -- ToAffine g1.go:322-340 / g2.go:365-386 + SerializeBytes) */
API void rc_g1_jac_to_affine_bytes_batch(const u64 *p, u8 *out, size_t n) { for (size_t i = 0; i < n; i++) rc_g1_jac_to_affine_bytes(p + 18 * i, out + 96 * i); }
API int rc_g2_jac_to_affine_bytes(const u64 p[36], u8 out[192]);
API void rc_g2_jac_to_affine_bytes_batch(const u64 *p, u8 *out, size_t n) { for (size_t i = 0; i < n; i++) rc_g2_jac_to_affine_bytes(p + 36 * i, out + 192 * i); }
API int rc_g2_jac_to_affine_bytes(const u64 p[36], u8 out[192]) { g2_jac a; g2_aff r; memcpy(&a, p, 288); g2_to_affine(&r, &a); if (r.inf) { memset(out, 0, 192); return 1; } g2_write_affine(out, &r); return 0; }

/* scalar multiplication: affine BE in, scalar 32-byte BE, affine BE out; returns 1 if result is infinity */
API int rc_g1_mul(const u8 p[96], const u8 k[32], u8 out[96]) {                                                                    /* g1.go:80-90 (MulFR) */
    g1_aff a, r; g1_jac j; u64 s[4];
    g1_read_affine(&a, p); repr_from_be(s, k, 4);
    g1_aff_mul(&j, &a, s, 4); g1_to_affine(&r, &j);
    if (r.inf) { memset(out, 0, 96); return 1; }
    g1_write_affine(out, &r); return 0;
}
API int rc_g2_mul(const u8 p[192], const u8 k[32], u8 out[192]) {                                                                  /* g2.go:92-102 */
    g2_aff a, r; g2_jac j; u64 s[4];
    g2_read_affine(&a, p); repr_from_be(s, k, 4);
    g2_aff_mul(&j, &a, s, 4); g2_to_affine(&r, &j);
    if (r.inf) { memset(out, 0, 192); return 1; }
    g2_write_affine(out, &r); return 0;
}
API void rc_g1_generator(u8 out[96]) { g1_aff g = g1_generator(); g1_write_affine(out, &g); }
API void rc_g2_generator(u8 out[192]) { g2_aff g = g2_generator(); g2_write_affine(out, &g); }
/* sequential Jacobian sums from the zero point (g2pubs/bls.go:165-192); inf_flags may be NULL */
API int rc_g1_sum(const u8 *pts, const u8 *inf_flags, size_t n, u8 out[96]) {
    g1_jac acc = g1_jac_zero(), t;
    for (size_t i = 0; i < n; i++) { g1_aff a; g1_jac j; g1_read_affine(&a, pts + 96 * i); if (inf_flags && inf_flags[i]) a.inf = 1; g1_to_jac(&j, &a); g1_add(&t, &acc, &j); acc = t; }
    g1_aff r; g1_to_affine(&r, &acc);
    if (r.inf) { memset(out, 0, 96); return 1; }
    g1_write_affine(out, &r); return 0;
}
API int rc_g2_sum(const u8 *pts, const u8 *inf_flags, size_t n, u8 out[192]) {
    g2_jac acc = g2_jac_zero(), t;
    for (size_t i = 0; i < n; i++) { g2_aff a; g2_jac j; g2_read_affine(&a, pts + 192 * i); if (inf_flags && inf_flags[i]) a.inf = 1; g2_to_jac(&j, &a); g2_add(&t, &acc, &j); acc = t; }
    g2_aff r; g2_to_affine(&r, &acc);
    if (r.inf) { memset(out, 0, 192); return 1; }
    g2_write_affine(out, &r); return 0;
}

API int rc_g1_compress(const u8 p[96], int inf, u8 out[48]) { g1_aff a; g1_read_affine(&a, p); a.inf = inf; g1_compress(out, &a); return 0; }
API int rc_g2_compress(const u8 p[192], int inf, u8 out[96]) { g2_aff a; g2_read_affine(&a, p); a.inf = inf; g2_compress(out, &a); return 0; }
/* returns error code (see above); *inf set for the infinity encoding; checked = also subgroup test (g1.go:185-198, g2.go:219-230) */
API int rc_g1_decompress(const u8 in[48], int checked, u8 out[96], int *inf) {
    g1_aff a; int e = g1_decompress_unchecked(&a, in);
    if (e) return e;
    *inf = a.inf;
    if (a.inf) { memset(out, 0, 96); return 0; }
    if (checked && !g1_in_subgroup(&a)) return 4;
    g1_write_affine(out, &a); return 0;
}
API int rc_g2_decompress(const u8 in[96], int checked, u8 out[192], int *inf) {
    g2_aff a; int e = g2_decompress_unchecked(&a, in);
    if (e) return e;
    *inf = a.inf;
    if (a.inf) { memset(out, 0, 192); return 0; }
    if (checked && !g2_in_subgroup(&a)) return 4;
    g2_write_affine(out, &a); return 0;
}

API void rc_hash_g1(const u8 *msg, size_t n, u8 out[96]) { g1_aff h; hash_g1(&h, msg, n); g1_write_affine(out, &h); }
API int rc_hash_g2(const u8 *msg, size_t n, u8 out[192]) { g2_aff h; if (!hash_g2(&h, msg, n)) return 0; g2_write_affine(out, &h); return 1; }
API void rc_hash_g2_with_domain(const u8 msg[32], const u8 domain[8], u8 out[192]) { g2_jac j; g2_aff a; hash_g2_with_domain(&j, msg, domain); g2_to_affine(&a, &j); g2_write_affine(out, &a); }

/* G2 prepare: 68 x 3 Fq2 Montgomery limbs (68*3*12 u64) */
API int rc_g2_prepare(const u8 q[192], u64 *out) {
    g2_aff a; g2_read_affine(&a, q);
    g2_prepared *p = (g2_prepared *)malloc(sizeof *p);
    g2_prepare(p, &a);
    memcpy(out, p->c, sizeof p->c);
    free(p);
    return N_COEFFS;
}
/* Miller loop over n pairs -> Fq12 Montgomery limbs (pre final exponentiation) */
API void rc_miller_loop(const u8 *g1s, const u8 *g2s, size_t n, u64 out[72]) {
    g1_aff *ps = (g1_aff *)malloc(n * sizeof(g1_aff));
    g2_prepared *qs = (g2_prepared *)malloc(n * sizeof(g2_prepared));
    for (size_t i = 0; i < n; i++) { g2_aff q; g1_read_affine(&ps[i], g1s + 96 * i); g2_read_affine(&q, g2s + 192 * i); g2_prepare(&qs[i], &q); }
    fq12 f; miller_loop(&f, ps, qs, (int)n);
    memcpy(out, &f, 576);
    free(ps); free(qs);
}
/* config 2: n independent reference Pairing() calls; out = n x 72 u64 Montgomery limbs */
API int rc_pairing_batch(const u8 *g1s, const u8 *g2s, u64 *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        g1_aff p; g2_aff q; fq12 e;
        g1_read_affine(&p, g1s + 96 * i); g2_read_affine(&q, g2s + 192 * i);
        if (!pairing_aff(&e, &p, &q)) return -1;
        memcpy(out + 72 * i, &e, 576);
    }
    return 0;
}

/* hash.go:9-39 HashSecretKey -> 32-byte BE normal-form scalar (512-bit value mod r by shift-subtract) */
API void rc_hash_secret_key(const u8 in[32], u8 out[32]) {
    u8 mp[33], t[64], tail[2];
    sha256_3(mp, in, 32, 0, 0, 0, 0); mp[32] = 0;
    for (int j = 1; j <= 2; j++) { tail[0] = 0x01; tail[1] = (u8)j; sha256_3(t + 32 * (j - 1), mp, 33, tail, 2, 0, 0); }
    u64 rem[5] = {0, 0, 0, 0, 0};
    for (int bit = 0; bit < 512; bit++) {
        int b = (t[bit / 8] >> (7 - bit % 8)) & 1;
        for (int i = 4; i > 0; i--) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
        rem[0] = (rem[0] << 1) | (u64)b;
        if (rem[4] || repr_cmp(rem, RC_RORDER, 4) >= 0) { u64 bw = 0; for (int i = 0; i < 4; i++) rem[i] = sbb(rem[i], RC_RORDER[i], &bw); rem[4] -= bw; }
    }
    repr_to_be(out, rem, 4);
}

/* ---- g2pubs (PublicKey in G2, Signature in G1, H: msg -> G1)  g2pubs/bls.go ---- */
API void rc_g2pubs_priv_to_pub(const u8 sk[32], u8 pk[192]) { u8 g[192]; rc_g2_generator(g); rc_g2_mul(g, sk, pk); }              /* :138-140 */
API void rc_g2pubs_sign(const u8 *msg, size_t n, const u8 sk[32], u8 sig[96]) { u8 h[96]; rc_hash_g1(msg, n, h); rc_g1_mul(h, sk, sig); }  /* :132-135 */
API int rc_g2pubs_verify(const u8 *msg, size_t n, const u8 pk[192], int pk_inf, const u8 sig[96], int sig_inf) {                   /* :159-162 */
    g1_aff h, s; g2_aff p, g = g2_generator();
    hash_g1(&h, msg, n);
    g1_read_affine(&s, sig); s.inf = sig_inf;
    g2_read_affine(&p, pk); p.inf = pk_inf;
    return compare_two_pairings(&s, &g, &h, &p);
}
static int has_duplicates(const u8 *msgs, const u64 *off, size_t n);
API int rc_g2pubs_verify_aggregate(const u8 *msgs, const u64 *off, const u8 *pks, const u8 sig[96], size_t n) {                    /* :240-270 */
    if (has_duplicates(msgs, off, n)) return 0;
    g1_aff s; g2_aff g = g2_generator(); fq12 lhs, rhs = fq12_one(), e;
    g1_read_affine(&s, sig);
    if (!pairing_aff(&lhs, &s, &g)) return 0;
    for (size_t i = 0; i < n; i++) {
        g1_aff h; g2_aff p;
        hash_g1(&h, msgs + off[i], off[i + 1] - off[i]);
        g2_read_affine(&p, pks + 192 * i);
        if (!pairing_aff(&e, &h, &p)) return 0;
        fq12_mul(&rhs, &rhs, &e);
    }
    return fq12_eq(&lhs, &rhs);
}
API int rc_g2pubs_verify_aggregate_common(const u8 *msg, size_t mlen, const u8 *pks, const u8 sig[96], size_t n) {                 /* :275-278 */
    u8 agg[192]; int inf = rc_g2_sum(pks, 0, n, agg);
    return rc_g2pubs_verify(msg, mlen, agg, inf, sig, 0);
}
/* ---- g1pubs (PublicKey in G1, Signature in G2, H: msg -> G2)  g1pubs/bls.go ---- */
API void rc_g1pubs_priv_to_pub(const u8 sk[32], u8 pk[96]) { u8 g[96]; rc_g1_generator(g); rc_g1_mul(g, sk, pk); }                /* :144-146 */
API void rc_g1pubs_sign(const u8 *msg, size_t n, const u8 sk[32], u8 sig[192]) { u8 h[192]; rc_hash_g2(msg, n, h); rc_g2_mul(h, sk, sig); } /* :132-135 */
API void rc_g1pubs_sign_with_domain(const u8 msg[32], const u8 sk[32], const u8 domain[8], u8 sig[192]) { u8 h[192]; rc_hash_g2_with_domain(msg, domain, h); rc_g2_mul(h, sk, sig); } /* :138-141 */
API int rc_g1pubs_verify(const u8 *msg, size_t n, const u8 pk[96], int pk_inf, const u8 sig[192], int sig_inf) {                   /* :165-168 */
    g2_aff h, s; g1_aff p, g = g1_generator();
    if (!hash_g2(&h, msg, n)) return 0;
    g2_read_affine(&s, sig); s.inf = sig_inf;
    g1_read_affine(&p, pk); p.inf = pk_inf;
    return compare_two_pairings(&g, &s, &p, &h);
}
API int rc_g1pubs_verify_with_domain(const u8 msg[32], const u8 pk[96], int pk_inf, const u8 sig[192], int sig_inf, const u8 domain[8]) {   /* :171-174 */
    g2_jac hj; g2_aff h, s; g1_aff p, g = g1_generator();
    hash_g2_with_domain(&hj, msg, domain); g2_to_affine(&h, &hj);
    g2_read_affine(&s, sig); s.inf = sig_inf;
    g1_read_affine(&p, pk); p.inf = pk_inf;
    return compare_two_pairings(&g, &s, &p, &h);
}
API int rc_g1pubs_verify_aggregate(const u8 *msgs, const u64 *off, const u8 *pks, const u8 sig[192], size_t n) {                   /* :252-282 */
    if (has_duplicates(msgs, off, n)) return 0;
    g2_aff s; g1_aff g = g1_generator(); fq12 lhs, rhs = fq12_one(), e;
    g2_read_affine(&s, sig);
    if (!pairing_aff(&lhs, &g, &s)) return 0;
    for (size_t i = 0; i < n; i++) {
        g2_aff h; g1_aff p;
        if (!hash_g2(&h, msgs + off[i], off[i + 1] - off[i])) return 0;
        g1_read_affine(&p, pks + 96 * i);
        if (!pairing_aff(&e, &p, &h)) return 0;
        fq12_mul(&rhs, &rhs, &e);
    }
    return fq12_eq(&lhs, &rhs);
}
API int rc_g1pubs_verify_aggregate_common(const u8 *msg, size_t mlen, const u8 *pks, const u8 sig[192], size_t n) {                /* :287-290 */
    u8 agg[96]; int inf = rc_g1_sum(pks, 0, n, agg);
    return rc_g1pubs_verify(msg, mlen, agg, inf, sig, 0);
}
API int rc_g1pubs_verify_aggregate_common_with_domain(const u8 msg[32], const u8 *pks, const u8 sig[192], size_t n, const u8 domain[8]) {  /* :294-297 */
    u8 agg[96]; int inf = rc_g1_sum(pks, 0, n, agg);
    return rc_g1pubs_verify_with_domain(msg, agg, inf, sig, 0, domain);
}
API int rc_g1pubs_verify_aggregate_with_domain(const u8 *msgs32, const u8 *pks, const u8 sig[192], size_t n, const u8 domain[8]) { /* :300-311 (no duplicate check) */
    g2_aff s; g1_aff g = g1_generator(); fq12 lhs, rhs = fq12_one(), e;
    g2_read_affine(&s, sig);
    if (!pairing_aff(&lhs, &g, &s)) return 0;
    for (size_t i = 0; i < n; i++) {
        g2_jac hj; g2_aff h; g1_aff p;
        hash_g2_with_domain(&hj, msgs32 + 32 * i, domain); g2_to_affine(&h, &hj);
        g1_read_affine(&p, pks + 96 * i);
        if (!pairing_aff(&e, &p, &h)) return 0;
        fq12_mul(&rhs, &rhs, &e);
    }
    return fq12_eq(&lhs, &rhs);
}

/* duplicate-message rejection (g2pubs/bls.go:245-261): sort copies bytewise, reject equal neighbours.
 * lastMsg starts as nil and bytes.Equal(m, nil) is true for an empty m, so an empty message in
 * first sorted position is rejected too -- reproduced here. */
typedef struct { const u8 *p; size_t n; } span;
static int span_cmp(const void *a, const void *b) {
    const span *x = (const span *)a, *y = (const span *)b;
    size_t m = x->n < y->n ? x->n : y->n;
    int c = m ? memcmp(x->p, y->p, m) : 0;
    if (c) return c;
    return (x->n > y->n) - (x->n < y->n);
}
static int has_duplicates(const u8 *msgs, const u64 *off, size_t n) {
    if (n == 0) return 0;
    span *s = (span *)malloc(n * sizeof(span));
    for (size_t i = 0; i < n; i++) { s[i].p = msgs + off[i]; s[i].n = off[i + 1] - off[i]; }
    qsort(s, n, sizeof(span), span_cmp);
    int dup = (s[0].n == 0);
    for (size_t i = 1; i < n && !dup; i++) dup = (span_cmp(&s[i - 1], &s[i]) == 0);
    free(s);
    return dup;
}
/* batch forms used as the checker for the product's batch C-ABI */
API void rc_g2pubs_verify_batch(const u8 *msgs, const u64 *off, const u8 *pks, const u8 *sigs, const u8 *inf_flags, u8 *ok, size_t n) {
    for (size_t i = 0; i < n; i++) {
        int pinf = inf_flags ? (inf_flags[i] & 1) : 0, sinf = inf_flags ? ((inf_flags[i] >> 1) & 1) : 0;
        ok[i] = (pinf || sinf) ? 0 : (u8)rc_g2pubs_verify(msgs + off[i], off[i + 1] - off[i], pks + 192 * i, 0, sigs + 96 * i, 0);
    }
}
API void rc_g1pubs_verify_batch(const u8 *msgs, const u64 *off, const u8 *pks, const u8 *sigs, const u8 *inf_flags, u8 *ok, size_t n) {
    for (size_t i = 0; i < n; i++) {
        int pinf = inf_flags ? (inf_flags[i] & 1) : 0, sinf = inf_flags ? ((inf_flags[i] >> 1) & 1) : 0;
        ok[i] = (pinf || sinf) ? 0 : (u8)rc_g1pubs_verify(msgs + off[i], off[i + 1] - off[i], pks + 96 * i, 0, sigs + 192 * i, 0);
    }
}
