// dpp_sub_probe.hip -- does `x - row_ror:14(x)` survive the compiler's DPP combine?  (round 6: the row layout's addition step computed
// P - rrot<7>(P) and got wrong values where the compiler had folded the move into v_subrev_u32_dpp vD, vX, vX.)
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_sub_probe.hip -o tools/dpp_sub_probe && tools/dpp_sub_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int* in, int* out) {
    const int x = in[threadIdx.x];
    const int r = __builtin_amdgcn_update_dpp(0, x, 0x12E, 0xf, 0xf, true);      // row_ror:14
    out[threadIdx.x] = x - r;                                                     // combine candidate
    int r2 = __builtin_amdgcn_update_dpp(0, x, 0x12E, 0xf, 0xf, true);
    asm volatile("" : "+v"(r2));                                                  // keeps the move apart
    out[64 + threadIdx.x] = x - r2;
    out[128 + threadIdx.x] = r - x;                                               // the other order
    int a, b, c; const int y = x + 0;
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %1 row_ror:14 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a) : "v"(x));          // what the row kernel got
    int x2 = x; asm volatile("v_mov_b32 %0, %1" : "=v"(x2) : "v"(x));
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 row_ror:14 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(b) : "v"(x), "v"(x2)); // two registers, same value
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %1 row_ror:14 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(c) : "v"(x));             // dpp(x) - x
    out[192 + threadIdx.x] = a; out[256 + threadIdx.x] = b; out[320 + threadIdx.x] = c; (void)y;
}
int main() {
    int h[64], *di, *dout, o[384];
    for (int i = 0; i < 64; i++) h[i] = 1000 * i + 7;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0, bad3 = 0;
    for (int i = 0; i < 64; i++) {
        const int src = (i & ~15) | ((i + 2) & 15), want = h[i] - h[src];
        bad1 += o[i] != want; bad2 += o[64 + i] != want; bad3 += o[128 + i] != -want;
    }
    int bad4 = 0, bad5 = 0, bad6 = 0;
    for (int i = 0; i < 64; i++) {
        const int src = (i & ~15) | ((i + 2) & 15), want = h[i] - h[src];
        bad4 += o[192 + i] != want; bad5 += o[256 + i] != want; bad6 += o[320 + i] != -want;
    }
    printf("asm v_subrev_u32_dpp d, x, x: %d wrong lanes; d, x, copy: %d wrong; v_sub_u32_dpp d, x, x: %d wrong\n", bad4, bad5, bad6);
    printf("x - dpp(x) combined: %d wrong lanes; kept apart: %d wrong; dpp(x) - x: %d wrong\n", bad1, bad2, bad3);
    for (int i = 0; i < 16; i++) printf("lane %2d x %6d asm subrev_dpp(x, x) %6d subrev_dpp(x, copy) %6d sub_dpp(x, x) %6d | ", i, h[i], o[192 + i], o[256 + i], o[320 + i]), printf("lane %2d want %6d combined %6d apart %6d\n", i, h[i] - h[(i + 2) & 15], o[i], o[64 + i]);
    return 0;
}
