// One field element per WAVE -- limb j in lane j -- against one per lane: the interleaved Montgomery product as 15 steps of
// (broadcast a_i, multiply-add, reduce the lowest limb, shift the row by one lane), ~13 instructions a step instead of 570 in sequence.
// For chains of dependent multiplications with nothing to run beside them: the 379-squaring exponentiation of a square root in a
// single HashG1 / Sign / Verify call.  Checks the power against fp_pow_core and times both.
// (row_mul / row_pow themselves live in bls_amd/csrc/fp_row.cuh.)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I bls_amd/csrc -o tools/ubench_rowmul tools/ubench_rowmul.hip
#include "pairing.cuh"
#include "device_io.cuh"
#include <cstdio>
using namespace blsmi;

// MODE 0: one element per lane (fp_pow_core); MODE 1: one element per wave
template <int MODE> __global__ void __launch_bounds__(64) k_pow(const i32* in, i32* out, int reps) {
    const int lane = threadIdx.x;
    if (MODE == 0) {
        vlimbs x;
        for (int i = 0; i < NL; i++) x[i] = in[blockIdx.x * NL + i];      // every lane the block's element (only lane 0 stores)
        for (int r = 0; r < reps; r++) x = fp_pow_core(x, C_QM3O4, BLSMI_QM3O4_BITS);
        FpS y; for (int i = 0; i < NL; i++) y.v[i] = x[i];
        const FpC c = fp_canon(y);
        if (lane == 0) for (int i = 0; i < NL; i++) out[blockIdx.x * NL + i] = c.v[i];
    } else {
        const i32 qv = lane < NL ? C_Q[lane] : 0;
        i32 a = lane < NL ? in[blockIdx.x * NL + lane] : 0;
        for (int r = 0; r < reps; r++) a = row_pow(a, C_QM3O4, BLSMI_QM3O4_BITS, qv);
        FpS y; for (int i = 0; i < NL; i++) y.v[i] = __builtin_amdgcn_readlane(a, i);
        const FpC c = fp_canon(y);
        if (lane == 0) for (int i = 0; i < NL; i++) out[blockIdx.x * NL + i] = c.v[i];
    }
}
int main() {
    const int nb = 64, reps = 4;
    i32 h[nb * NL];
    unsigned s = 12345;
    for (int i = 0; i < nb * NL; i++) { s = s * 1664525u + 1013904223u; h[i] = (i32)(s >> 5) & MASK; }
    for (int b = 0; b < nb; b++) h[b * NL + NL - 1] &= 0x3ff;              // below 2^388: a few q, fine for Montgomery inputs
    i32 *din, *d0, *d1; hipMalloc(&din, sizeof h); hipMalloc(&d0, sizeof h); hipMalloc(&d1, sizeof h);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms0 = 0, ms1 = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_pow<0>, dim3(nb), dim3(64), 0, 0, din, d0, reps); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms0, e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(k_pow<1>, dim3(nb), dim3(64), 0, 0, din, d1, reps); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms1, e0, e1);
    }
    i32 r0[nb * NL], r1[nb * NL];
    hipMemcpy(r0, d0, sizeof r0, hipMemcpyDeviceToHost); hipMemcpy(r1, d1, sizeof r1, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < nb * NL; i++) bad += r0[i] != r1[i];
    printf("x^((q-3)/4), %d in a row, %d elements: one per lane %.3f ms (%.1f us per power), one per wave %.3f ms (%.1f us per power); mismatching limbs: %d\n",
           reps, nb, ms0, ms0 * 1e3 / reps, ms1, ms1 * 1e3 / reps, bad);
    return bad != 0;
}
