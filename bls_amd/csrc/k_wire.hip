// k_wire.hip -- the compressed wire format (DecompressG1/G2, CompressG1/G2, subgroup checks), verdict / flag housekeeping and the Fq / Fq2-level
// unit ops of the parity tests (split from k_hash.hip: the two halves compile in parallel).
#include "hash.cuh"
#include "device_io.cuh"

// ---- compressed wire format (g1.go:185-249, g2.go:219-295) ----------------------------------------
// IsInCorrectSubgroupAssumingOnCurve (g1.go:137-141, g2.go:293-295) tests r * P == infinity with a 255-bit
// double-and-add.  The same predicate, for a point ON the curve, through the curve's endomorphisms:
//   G1: phi(x, y) = (beta x, y) satisfies phi^2 + phi + 1 = 0; phi(P) = [-x^2] P implies (x^4 - x^2 + 1) P = r P = 0,
//       and on G1 phi acts as that eigenvalue: two 64-bit multiplications instead of one 255-bit one.
//   G2: psi (untwist-Frobenius-twist) satisfies psi^2 - t psi + q = 0 with t = x + 1; psi(P) = [x] P implies
//       (q - x) P = (h1 r) P = 0, so the order of P divides r gcd(h1, h2) = r (the cofactors are coprime), and on
//       G2 psi acts as x: one 64-bit multiplication.
// The slow form is kept (in_subgroup_by_order) and the two are compared on curve points inside and outside
// the subgroup by tests/test_gpu_verify.py::test_wire_format.
template <class F> BLSMI_DEV bool in_subgroup_by_order(const Aff<F>& p) {
    Jac<F> res = to_jac(p);
    for (int i = BLSMI_R_ORDER_BITS - 2; i >= 0; i--) {
        res = jac_double(res);
        if ((C_R_ORDER[i >> 5] >> (i & 31)) & 1) res = jac_add_affine(res, p);
    }
    return res.inf != 0;
}
BLSMI_DEV bool in_subgroup(const G1Aff& p) {                             // [x^2] P + phi(P) == infinity
    const G1Jac j = jac_mul_u64_public(aff_mul_u64_public(p, BLSMI_X_ABS), BLSMI_X_ABS);
    G1Aff ph; ph.x = fp_store(fp_mul(p.x, C_BETA)); ph.y = p.y; ph.inf = 0;
    return jac_add_affine(j, ph).inf != 0;
}
BLSMI_DEV bool in_subgroup(const G2Aff& p) {                             // [|x|] P + psi(P) == infinity  (x < 0)
    const G2Jac j = aff_mul_u64_public(p, BLSMI_X_ABS);
    G2Aff ps; psi(ps, p);
    return jac_add_affine(j, ps).inf != 0;
}
BLSMI_DEV FpS load_be48_masked(const u8* p) {                            // clears the three flag bits of the first byte
    const u32* w32 = reinterpret_cast<const u32*>(p);
    u32 w[12];
#pragma unroll
    for (int j = 0; j < 12; j++) w[j] = __builtin_bswap32(w32[11 - j]);
    w[11] &= 0x1fffffffu;
    return fp_from_words(w);
}
// WAVE: one point per 64-lane workgroup (t = blockIdx.x, every lane the same values, lane 0 stores), the square root's exponentiation
// with one limb per lane (fp_row.cuh) -- the smallest calls; no subgroup test in that form (the caller runs it as a level program)
template <bool WAVE>
BLSMI_DEV void g1_decompress_body(const u8* in, int check, u8* out, u8* out_inf, u8* err, size_t n) {
    const size_t t = WAVE ? (size_t)blockIdx.x : (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    const u8* c = in + 48 * tt;
    const u8 b0 = c[0];
    u32 rest = b0 & 0x3f;
    for (int i = 1; i < 48; i++) rest |= c[i];
    const FpS x = load_be48_masked(c);
    bool ok;
    const FpS y = fp_sqrt<WAVE>(fp_add(fp_mul(fp_sqr(x), x), C_B), ok);  // g1.go:111-132
    const i32 lt = ~fp_gt_half(y);                                         // y < -y
    const i32 greatest = (b0 & 0x20) ? -1 : 0;
    G1Aff a; a.x = x; a.y = fp_select(lt ^ greatest, y, fp_store(fp_neg(y))); a.inf = 0;
    const bool sub = check == 2 ? in_subgroup_by_order(a) : check ? in_subgroup(a) : true;   // check 2: the reference's r * P form
    u8 e = 0; u8 inf = 0;
    if (!(b0 & 0x80)) e = 1;
    else if (b0 & 0x40) { if (rest) e = 2; else inf = 1; }
    else if (!ok) e = 3;
    else if (!sub) e = 4;
    if (t < n && (!WAVE || threadIdx.x == 0)) {
        a.inf = (inf || e) ? -1 : 0;
        store_g1(out + 96 * t, a);
        out_inf[t] = inf; err[t] = e;
    }
}
KERNEL2 k_g1_decompress(const u8* in, int check, u8* out, u8* out_inf, u8* err, size_t n) { g1_decompress_body<false>(in, check, out, out_inf, err, n); }
__global__ void __launch_bounds__(64, 2) k_g1_decompress_waves(const u8* in, u8* out, u8* out_inf, u8* err, size_t n) { g1_decompress_body<true>(in, 0, out, out_inf, err, n); }
template <bool WAVE>
BLSMI_DEV void g2_decompress_body(const u8* in, int check, u8* out, u8* out_inf, u8* err, size_t n) {
    const size_t t = WAVE ? (size_t)blockIdx.x : (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    const u8* c = in + 96 * tt;
    const u8 b0 = c[0];
    u32 rest = b0 & 0x3f;
    for (int i = 1; i < 96; i++) rest |= c[i];
    Fp2S x; x.c1 = load_be48_masked(c); x.c0 = load_be48(c + 48);          // x.c1 || x.c0 on the wire (g2.go:254-255)
    bool ok;
    const Fp2S y = fp2_sqrt_any<WAVE>(fp2_add(fp2_mul(fp2_sqr(x), x), C_B2), ok);   // g2.go:149-169; y or -y is chosen below
    const i32 lt = ~fp2_sign_is_neg(y);
    const i32 greatest = (b0 & 0x20) ? -1 : 0;
    G2Aff a; a.x = x; a.y = fp2_select(lt ^ greatest, y, fp2_store(fp2_neg(y))); a.inf = 0;
    const bool sub = check == 2 ? in_subgroup_by_order(a) : check ? in_subgroup(a) : true;
    u8 e = 0; u8 inf = 0;
    if (!(b0 & 0x80)) e = 1;
    else if (b0 & 0x40) { if (rest) e = 2; else inf = 1; }
    else if (!ok) e = 3;
    else if (!sub) e = 4;
    if (t < n && (!WAVE || threadIdx.x == 0)) {
        a.inf = (inf || e) ? -1 : 0;
        store_g2(out + 192 * t, a);
        out_inf[t] = inf; err[t] = e;
    }
}
KERNEL k_g2_decompress(const u8* in, int check, u8* out, u8* out_inf, u8* err, size_t n) { g2_decompress_body<false>(in, check, out, out_inf, err, n); }
__global__ void __launch_bounds__(64) k_g2_decompress_waves(const u8* in, u8* out, u8* out_inf, u8* err, size_t n) { g2_decompress_body<true>(in, 0, out, out_inf, err, n); }
// Small batches: the decompression kernels run without their subgroup test, the test runs as a level program of the latency
// path (k_lat.hip: subgrp1 / subgrp2, one point per wave) and this kernel applies its verdict -- what the kernels above do
// for e = 4: the record becomes the all-zero (infinity) record and the error code is set.
KERNEL k_apply_subgroup(const u8* in_subgroup, u8* out, int rec_words, const u8* out_inf, u8* err, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n || err[t] || out_inf[t] || in_subgroup[t]) return;
    err[t] = 4;
    u32* w = reinterpret_cast<u32*>(out) + (size_t)rec_words * t;
    for (int i = 0; i < rec_words; i++) w[i] = 0;
}
// tuple flags for the verify kernels from the two deserialisation results: bit 0 = unusable public key, bit 1 = unusable signature
KERNEL k_merge_flags(const u8* inf_pk, const u8* err_pk, const u8* inf_sig, const u8* err_sig, u8* flags, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t < n) flags[t] = (u8)(((inf_pk[t] | err_pk[t]) ? 1 : 0) | ((inf_sig[t] | err_sig[t]) ? 2 : 0));
}
// tuple flags of a verify batch: the caller's flags (may be null) OR-ed with "the key / signature record is all zero", the
// library's encoding of the point at infinity; *any (may be null) is raised when some tuple is flagged
KERNEL k_flag_zero_records(const u8* pks, int pk_words, const u8* sigs, int sig_words, const u8* in_flags, u8* flags, int* any, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    u32 a = 0, b = 1;
    const u32* p = reinterpret_cast<const u32*>(pks) + (size_t)pk_words * t;
    for (int i = 0; i < pk_words; i++) a |= p[i];
    if (sigs) {
        b = 0;
        const u32* q = reinterpret_cast<const u32*>(sigs) + (size_t)sig_words * t;
        for (int i = 0; i < sig_words; i++) b |= q[i];
    }
    const u8 f = (u8)((in_flags ? in_flags[t] : 0) | (a ? 0 : 1) | (b ? 0 : 2));
    flags[t] = f;
    if (f && any) atomicOr(any, 1);
}
// verdict bytes -> bits, LSB first: bitmap[b] holds tuples 8b .. 8b+7 (the layout of the bitmap all-reduce; `bitmap` points
// at this shard's first byte, shards start on multiples of 8 tuples)
// an int32 infinity flag (the sums' verdict) as the flag byte of a one-tuple verify
KERNEL k_flag_to_byte(const i32* flag, u8* out) { if (blockIdx.x == 0 && threadIdx.x == 0) *out = *flag ? 1 : 0; }
KERNEL k_pack_bitmap(const u8* ok, u8* bitmap, size_t n) {
    const size_t b = (size_t)blockIdx.x * WG + threadIdx.x;
    if (8 * b >= n) return;
    u32 v = 0;
    for (int i = 0; i < 8; i++) if (8 * b + i < n && ok[8 * b + i]) v |= 1u << i;
    bitmap[b] = (u8)v;
}
KERNEL k_g1_compress(const u8* pts, const u8* in_inf, u8* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    u8* o = out + 48 * t;
    if (in_inf && in_inf[t]) { for (int i = 0; i < 48; i++) o[i] = 0; o[0] = 0xc0; return; }
    const u8* p = pts + 96 * t;
    for (int i = 0; i < 48; i++) o[i] = p[i];
    const FpS y = load_be48(p + 48);
    o[0] |= 0x80 | (fp_gt_half(y) ? 0x20 : 0);                             // g1.go:239-246
}
KERNEL k_g2_compress(const u8* pts, const u8* in_inf, u8* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    u8* o = out + 96 * t;
    if (in_inf && in_inf[t]) { for (int i = 0; i < 96; i++) o[i] = 0; o[0] = 0xc0; return; }
    const u8* p = pts + 192 * t;
    for (int i = 0; i < 48; i++) { o[i] = p[48 + i]; o[48 + i] = p[i]; }    // x.c1 || x.c0 (g2.go:273-276)
    Fp2S y; y.c0 = load_be48(p + 96); y.c1 = load_be48(p + 144);
    o[0] |= 0x80 | (fp2_sign_is_neg(y) ? 0x20 : 0);                        // g2.go:278-284
}

// ---- the reference's in-memory points at the boundary (blsmi 0.6: the *_jac entry points) ---------------------------------------
// A Go caller holds *bls.G1Projective / *bls.G2Projective (g2pubs/bls.go:13-15, 53-55): Jacobian coordinates, each FQ 6 LE u64
// Montgomery(2^384) limbs.  Taking them as they lie saves the shim one ToAffine (an Fq inversion, g1.go:322-340) and one
// SerializeBytes (two MontReduce + byte swap, g1.go:157-167) per point on a host core -- ~12 us, against 0.36 us of device time
// per verify.  This kernel is that ToAffine + SerializeBytes: n records of 18 / 36 u64 -> n wire records (the all-zero record for
// z == 0, which every kernel downstream reads as infinity) + the infinity flags.  A wave whose 64 points all have z == 1
// (keys and signatures that came through Deserialize*) skips the inversion, like the reference's shortcut g2.go:368.
template <class F, int PB>
BLSMI_DEV void jac_to_affine_body(const u64* in, u8* out, u8* out_inf, size_t n) {
    constexpr int NC = sizeof(F) / sizeof(FpS);
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    bool z_one;
    const Jac<F> j = load_jac_m384<F>(in + (size_t)18 * NC * tt, &z_one);
    Aff<F> a;
    if (__all(z_one || j.inf)) { a.x = j.x; a.y = j.y; a.inf = j.inf; }
    else a = jac_to_affine(j);
    if (t < n) {
        if constexpr (NC == 1) store_g1(out + (size_t)PB * t, a); else store_g2(out + (size_t)PB * t, a);
        if (out_inf) out_inf[t] = a.inf ? 1 : 0;
    }
}
KERNEL2 k_g1_jac_to_affine(const u64* in, u8* out, u8* out_inf, size_t n) { jac_to_affine_body<FpS, 96>(in, out, out_inf, n); }
KERNEL k_g2_jac_to_affine(const u64* in, u8* out, u8* out_inf, size_t n) { jac_to_affine_body<Fp2S, 192>(in, out, out_inf, n); }
// the other direction, for results that go back into a Go value (AggregatePublicKeys / AggregateSignatures: one point): wire record
// + infinity flag -> the in-memory record with z = 1
// (in_inf: n flags, int32 -- the sums' -- or, inf_u8 != 0, bytes -- the multiplication kernels')
KERNEL k_affine_to_jac(const u8* in, const void* in_inf, int inf_u8, int group, u64* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    const bool inf = in_inf && (inf_u8 ? static_cast<const u8*>(in_inf)[t] != 0 : static_cast<const i32*>(in_inf)[t] != 0);
    if (group == 1) { G1Aff a = load_g1(in + 96 * t); if (inf) a.inf = -1; store_jac_m384(out + 18 * t, a); }
    else { G2Aff a = load_g2(in + 192 * t); if (inf) a.inf = -1; store_jac_m384(out + 36 * t, a); }
}

KERNEL k_debug_fq(int op, const u64* a, const u64* b, u64* out, u8* flag, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    const FpS x = load_m384(a + 6 * t);
    FpS y = fp_zero();
    if (op == BLSMI_OP_FQ_MUL || op == BLSMI_OP_FQ_ADD || op == BLSMI_OP_FQ_SUB || op == BLSMI_OP_FQ_CMP) y = load_m384(b + 6 * t);
    FpS r = fp_zero();
    bool ok = true;
    int code = -1;
    switch (op) {
        case BLSMI_OP_FQ_DBL: r = fp_store(fp_dbl(x)); break;                                       // fq.go:140-143
        case BLSMI_OP_FQ_CMP: {                                                                      // fq.go:134-137: order of the normal forms
            u32 wx[12], wy[12];
            fp_to_words(x, wx); fp_to_words(y, wy);
            int c = 0;
            for (int j = 0; j < 12; j++) if (wx[j] != wy[j]) c = wx[j] > wy[j] ? 1 : -1;           // most significant difference wins
            code = c + 1; r = x; break;
        }
        case BLSMI_OP_FQ_PARITY: code = fp_gt_half(x) ? 1 : 0; r = x; break;                        // fq.go:269-273: a > -a
        case BLSMI_OP_FQ_MUL: r = fp_store(fp_mul(x, y)); break;
        case BLSMI_OP_FQ_SQR: r = fp_store(fp_sqr(x)); break;
        case BLSMI_OP_FQ_ADD: r = fp_store(fp_add(x, y)); break;
        case BLSMI_OP_FQ_SUB: r = fp_store(fp_sub(x, y)); break;
        case BLSMI_OP_FQ_NEG: r = fp_store(fp_neg(x)); break;
        case BLSMI_OP_FQ_INV: r = fp_inv(x); ok = !fp_is_zero(x); break;
        case BLSMI_OP_FQ_SQRT: r = fp_sqrt(x, ok); break;
    }
    store_m384(out + 6 * t, r);
    if (flag) flag[t] = code >= 0 ? (u8)code : (ok ? 1 : 0);
}
KERNEL k_debug_fq2(int op, const u64* a, const u64* b, u64* out, u8* flag, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<2> ra = rec_load<2>(a, t), rb = ra, ro;
    if (op == BLSMI_OP_FQ2_MUL) rb = rec_load<2>(b, t);
    const Fp2S x = as<Fp2S>(ra), y = as<Fp2S>(rb);
    Fp2S r = fp2_zero();
    bool ok = true;
    switch (op) {
        case BLSMI_OP_FQ2_MUL: r = fp2_store(fp2_mul(x, y)); break;
        case BLSMI_OP_FQ2_SQR: r = fp2_store(fp2_sqr(x)); break;
        case BLSMI_OP_FQ2_INV: r = fp2_store(fp2_inv(x)); ok = !fp2_is_zero(x); break;
        case BLSMI_OP_FQ2_MUL_NR: r = fp2_store(fp2_mul_nr(x)); break;
        case BLSMI_OP_FQ2_SQRT: r = fp2_sqrt(x, ok); break;
        case BLSMI_OP_FQ2_SQRT_ANY: r = fp2_sqrt_any(x, ok); break;
        case BLSMI_OP_FQ2_PARITY: ok = fp2_sign_is_neg(x) != 0; r = x; break;                        // fq2.go:256-260: a > -a, c1 first
    }
    as<Fp2S>(ro) = r;
    rec_store<2>(out, t, ro);
    if (flag) flag[t] = ok ? 1 : 0;
}
// SWU helpers on a caller-chosen t (own kernels: the G1 helper keeps the two-waves-per-SIMD register budget of k_hash_g1)
KERNEL2 k_debug_swu_g1(const u64* a, u64* out, size_t n) {               // optimizedSWUMapHelper (g1.go:628-714)
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<3> r = rec_load<3>(a, t);
    G1Aff p; swu_g1_helper(p, reinterpret_cast<FpS*>(&r)[0]);
    reinterpret_cast<FpS*>(&r)[0] = p.x; reinterpret_cast<FpS*>(&r)[1] = p.y; reinterpret_cast<FpS*>(&r)[2] = fp_zero();
    rec_store<3>(out, t, r);
}
KERNEL k_debug_swu_g2(const u64* a, u64* out, size_t n) {                // OptimizedSWU2MapHelper (g2.go:933-1031)
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<6> r = rec_load<6>(a, t);
    G2Aff p; swu_g2_helper(p, reinterpret_cast<Fp2S*>(&r)[0]);
    reinterpret_cast<Fp2S*>(&r)[0] = p.x; reinterpret_cast<Fp2S*>(&r)[1] = p.y; reinterpret_cast<Fp2S*>(&r)[2] = fp2_zero();
    rec_store<6>(out, t, r);
}

