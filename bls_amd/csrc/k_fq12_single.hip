// k_fq12_single.hip -- one-tuple-per-lane layout: the Fq12 product tree of VerifyAggregate, format conversions and the Fq12-level unit ops
// of the parity tests (split from k_fe_single.hip so that the from-scratch build's longest unit is shorter: the two halves compile in parallel).
#include "pairing.cuh"
#include "device_io.cuh"

KERNEL k_fq12_from_m384(const u64* in, i32* fbuf, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    for (int e = 0; e < 12; e++) soa_store(fbuf, n, t, e, load_m384(in + 72 * t + 6 * e));
}

// Fq12 product tree for VerifyAggregate: dst[t] = src[t] * src[t + half]
KERNEL k_fq12_prod_level(const i32* src, i32* dst, size_t n, size_t half) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= half) return;
    Fp12S a = soa_load12(src, n, t);
    if (t + half < n) { const Fp12S b = soa_load12(src, n, t + half); nf_fp12_mul(a, a, b); }
    soa_store12(dst, half, t, a);
}
// n Fq12 values stored record after record (180 words each: what the all-gather of per-device partial products delivers)
// -> the structure-of-arrays layout of the product tree
KERNEL k_fq12_aos_to_soa(const i32* aos, i32* soa, size_t n) {
    const size_t i = (size_t)blockIdx.x * WG + threadIdx.x;
    if (i >= 180 * n) return;
    const size_t t = i / 180, w = i % 180;
    soa[w * n + t] = aos[i];
}
KERNEL k_fq12_one(i32* f) { if (blockIdx.x == 0 && threadIdx.x == 0) soa_store12(f, 1, 0, fp12_one()); }
KERNEL k_debug_fq12(int op, const u64* a, const u64* b, u64* out, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    if (t >= n) return;
    Rec<12> ra = rec_load<12>(a, t), rb = ra, ro;
    if (op == BLSMI_OP_FQ12_MUL || op == BLSMI_OP_FQ12_MUL_BY_014 || op == BLSMI_OP_FQ12_MUL_BY_LINE_PAIR) rb = rec_load<12>(b, t);
    const Fp12S x = as<Fp12S>(ra), y = as<Fp12S>(rb);
    Fp12S r = fp12_one();
    switch (op) {
        case BLSMI_OP_FQ12_MUL_BY_014: r = fp12_store(fp12_mul_by_014(x, y.c0.c0, y.c0.c1, y.c0.c2)); break;   // fq12.go:32-47, (c0, c1, c4) = first three Fq2 of b
        case BLSMI_OP_FQ12_MUL_BY_LINE_PAIR: {                                                       // two 014 elements at once (ell2 of the 2-pair Miller loop)
            const LinePair m = line_pair_product(y.c0.c0, y.c0.c1, y.c0.c2, y.c1.c0, y.c1.c1, y.c1.c2);
            r = fp12_store(fp12_mul_by_line_pair(x, m)); break;
        }
        case BLSMI_OP_FQ12_MUL: r = fp12_store(fp12_mul(x, y)); break;
        case BLSMI_OP_FQ12_SQR: r = fp12_store(fp12_sqr(x)); break;
        case BLSMI_OP_FQ12_INV: r = fp12_store(fp12_inv(x)); break;
        case BLSMI_OP_FQ12_FROB1: r = fp12_store(fp12_frob<1>(x)); break;
        case BLSMI_OP_FQ12_FROB2: r = fp12_store(fp12_frob<2>(x)); break;
        case BLSMI_OP_FQ12_FROB3: r = fp12_store(fp12_frob<3>(x)); break;
        case BLSMI_OP_FQ12_CYCLO_SQR: r = fp12_cyclotomic_sqr(x); break;
        case BLSMI_OP_FQ12_CYCLO_RUN16: r = cyc_sqr_run(x, 16); break;
    }
    as<Fp12S>(ro) = r;
    rec_store<12>(out, t, ro);
}
