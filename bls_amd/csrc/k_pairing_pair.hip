// k_pairing_pair.hip -- the default pairing kernels: lane pair per tuple (fp2_pair.inc), two waves per SIMD.
#include "pairing.cuh"
#include "device_io.cuh"

#define BLSMI_PAIR_MILLER_ONLY          // the final-exponentiation kernels of pair_kernels.inc are compiled in k_fe_pair.hip
#include "pair_kernels.inc"

// The same Fq2 / Fq6 / Fq12 operations in the LANE-PAIR layout (fp2_pair.inc): lanes 2k, 2k+1 hold the c0 / c1 halves of
// tuple k's W/2 Fq2 coefficients.  Ops that read a second operand take it from b.
template <int W2>
__device__ void pair_rec_load(FpS* c, const u64* p, size_t t, int par) { for (int j = 0; j < W2; j++) c[j] = load_m384(p + (size_t)6 * (2 * W2 * t + 2 * j + par)); }
template <int W2>
__device__ void pair_rec_store(u64* p, size_t t, int par, const FpS* c) { for (int j = 0; j < W2; j++) store_m384(p + (size_t)6 * (2 * W2 * t + 2 * j + par), c[j]); }
__global__ void __launch_bounds__(WG, BLSMI_PAIR_WAVES) k_debug_pairl(int op, const u64* a, const u64* b, u64* out, size_t n) {
    namespace P2 = blsmi::pairl;
    const int par = threadIdx.x & 1;
    const size_t t0 = (size_t)blockIdx.x * (WG / 2) + (threadIdx.x >> 1);
    const size_t t = t0 < n ? t0 : n - 1;                                 // both lanes of a pair stay active (DPP partner exchange)
    if (op < 32) {
        P2::Fp2S x, y, r; pair_rec_load<1>(&x.c, a, t, par); y = x; if (b) pair_rec_load<1>(&y.c, b, t, par);
        switch (op) {
            case BLSMI_OP_FQ2_MUL: r = P2::fp2_store(P2::fp2_mul(x, y)); break;
            case BLSMI_OP_FQ2_SQR: r = P2::fp2_store(P2::fp2_sqr(x)); break;
            case BLSMI_OP_FQ2_INV: r = P2::fp2_store(P2::fp2_inv(x)); break;
            default: r = P2::fp2_store(P2::fp2_mul_nr(x)); break;
        }
        if (t0 < n) pair_rec_store<1>(out, t, par, &r.c);
    } else if (op < 48) {
        P2::Fp6S x, y, r; pair_rec_load<3>(reinterpret_cast<FpS*>(&x), a, t, par); y = x; if (b) pair_rec_load<3>(reinterpret_cast<FpS*>(&y), b, t, par);
        switch (op) {
            case BLSMI_OP_FQ6_MUL: r = P2::fp6_store(P2::fp6_mul(x, y)); break;
            case BLSMI_OP_FQ6_SQR: r = P2::fp6_store(P2::fp6_sqr(x)); break;
            case BLSMI_OP_FQ6_INV: r = P2::fp6_store(P2::fp6_inv(x)); break;
            case BLSMI_OP_FQ6_MUL_BY_1: r = P2::fp6_store(P2::fp6_mul_by_1(x, y.c0)); break;
            case BLSMI_OP_FQ6_MUL_BY_01: r = P2::fp6_store(P2::fp6_mul_by_01(x, y.c0, y.c1)); break;
            default: r = P2::fp6_store(P2::fp6_frob<1>(x)); break;
        }
        if (t0 < n) pair_rec_store<3>(out, t, par, reinterpret_cast<const FpS*>(&r));
    } else {
        P2::Fp12S x, y, r; pair_rec_load<6>(reinterpret_cast<FpS*>(&x), a, t, par); y = x; if (b) pair_rec_load<6>(reinterpret_cast<FpS*>(&y), b, t, par);
        switch (op) {
            case BLSMI_OP_FQ12_MUL: P2::nf_fp12_mul(r, x, y); break;
            case BLSMI_OP_FQ12_SQR: P2::nf_fp12_sqr(r, x); break;
            case BLSMI_OP_FQ12_INV: P2::nf_fp12_inv(r, x); break;
            case BLSMI_OP_FQ12_FROB1: P2::nf_fp12_frob1(r, x); break;
            case BLSMI_OP_FQ12_FROB2: P2::nf_fp12_frob2(r, x); break;
            case BLSMI_OP_FQ12_FROB3: P2::nf_fp12_frob3(r, x); break;
            case BLSMI_OP_FQ12_CYCLO_SQR: P2::nf_fp12_cyc_sqr(r, x); break;
            case BLSMI_OP_FQ12_CYCLO_RUN16: r = P2::cyc_sqr_run(x, 16); break;
            case BLSMI_OP_FQ12_MUL_BY_014: r = P2::fp12_store(P2::fp12_mul_by_014(x, y.c0.c0, y.c0.c1, y.c0.c2)); break;
            default: {
                const P2::LinePair m = P2::line_pair_product(y.c0.c0, y.c0.c1, y.c0.c2, y.c1.c0, y.c1.c1, y.c1.c2);
                r = P2::fp12_store(P2::fp12_mul_by_line_pair(x, m)); break;
            }
        }
        if (t0 < n) pair_rec_store<6>(out, t, par, reinterpret_cast<const FpS*>(&r));
    }
}
__global__ void __launch_bounds__(WG, 2) k_debug_prepare_pair(const u8* g2, i32* table) {
    namespace P2 = blsmi::pairl;
    if (threadIdx.x >= 2) return;
    const int par = threadIdx.x & 1;
    P2::prepare_lines(P2::wrap(load_be48(g2 + 48 * par)), P2::wrap(load_be48(g2 + 96 + 48 * par)), table);
    for (int e = 0; e < 68 * 3; e++) P2::fp2_table_to_limbs27(table, e);      // the reader (k_debug_lines_to_m384) is a 27-bit-limb kernel
}
// G2AffineToPrepared of the G2 generator in THIS translation unit's own limbs (the table k_miller2_pair / k_miller2_quad read as `pre`:
// with 28-bit limbs here, the start-up table of k_prepare_generator_lines -- 27-bit limbs -- is not theirs to read)
__global__ void __launch_bounds__(WG, 2) k_prepare_generator_lines_pair(const u8* g2, i32* table) {
    namespace P2 = blsmi::pairl;
    if (threadIdx.x >= 2) return;
    const int par = threadIdx.x & 1;
    P2::prepare_lines(P2::wrap(load_be48(g2 + 48 * par)), P2::wrap(load_be48(g2 + 96 + 48 * par)), table);
}
