// k_fe_single.hip -- one-tuple-per-lane layout: final exponentiation kernels (the Fq12 product tree and the unit ops: k_fq12_single.hip).
#include "pairing.cuh"
#include "device_io.cuh"

// mode 0: out = FE(f) as Montgomery-384 limbs; mode 1: out = f itself (no final exponentiation)
KERNEL k_final_exp(const i32* fbuf, u64* out, size_t n, int mode) {
    __shared__ u32 lds[WG * 145];
    const size_t first = (size_t)blockIdx.x * WG;
    const size_t t = first + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    Fp12S f = soa_load12(fbuf, n, tt);
    if (mode == 0) final_exponentiation(f);
    const FpS* c = reinterpret_cast<const FpS*>(&f);
    for (int e = 0; e < 12; e++) {                                       // this lane's 576-byte record, into LDS
        u32 w[12];
        fp_to_mont384_words(c[e], w);
#pragma unroll
        for (int j = 0; j < 12; j++) lds[threadIdx.x * 145 + 12 * e + j] = w[j];
    }
    tile_store<144>(lds, reinterpret_cast<u8*>(out), first, n);         // coalesced write-out
}
// ok[t] = FinalExponentiation(f_t) == 1, and 0 for tuples flagged as containing a point at infinity
KERNEL k_final_exp_is_one(const i32* fbuf, const u8* inf_flags, u8* ok, size_t n) {
    const size_t t = (size_t)blockIdx.x * WG + threadIdx.x;
    const size_t tt = t < n ? t : n - 1;
    Fp12S f = soa_load12(fbuf, n, tt);
    final_exponentiation(f);
    const bool one = fp12_eq(f, fp12_one());
    if (t < n) ok[t] = (one && !(inf_flags && inf_flags[t])) ? 1 : 0;
}
// two-element compare for VerifyAggregate: ok = FE(a) == FE(b)
KERNEL k_final_exp_equal(const i32* a, const i32* b, i32* ok) {
    if (blockIdx.x != 0) return;
    Fp12S x = soa_load12(a, 1, 0), y = soa_load12(b, 1, 0);
    final_exponentiation(x);
    final_exponentiation(y);
    const bool e = fp12_eq(x, y);
    if (threadIdx.x == 0) *ok = e ? 1 : 0;
}
