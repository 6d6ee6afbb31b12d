// dpp_sub_probe2.hip -- the same question as dpp_sub_probe.hip for quad_perm controls: k_hash_quad.hip's homogeneous addition got `v_subrev_u32_dpp d, a, b quad_perm:[0,0,0,0]`
// from the compiler (b - quad_perm(a)) and the hash stayed bit-exact.  Is v_subrev_u32_dpp right with quad_perm and wrong with row_ror, or right with two different registers?
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_sub_probe2.hip -o tools/dpp_sub_probe2 && tools/dpp_sub_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const int* in, int* out) {
    const int x = in[threadIdx.x], y = in[64 + threadIdx.x];
    int a, b, c, d, e, f;
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a) : "v"(x), "v"(y));   // ISA: y - qp(x)
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(b) : "v"(x), "v"(y));      // ISA: qp(x) - y
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 row_ror:14 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(c) : "v"(x), "v"(y));            // ISA: y - ror(x)
    asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 row_ror:14 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(d) : "v"(x), "v"(y));               // ISA: ror(x) - y
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(e) : "v"(x), "v"(y));       // ISA: y - bcast3(x)
    asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f) : "v"(x), "v"(y));  // ISA: y - qswap(x)
    // what the compiler makes of the same expressions (look at the ISA: hipcc -S)
    const int qx = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xf, 0xf, true);       // quad_perm [0,0,0,0]
    const int rx = __builtin_amdgcn_update_dpp(0, x, 0x12E, 0xf, 0xf, true);      // row_ror:14
    int* o = out + threadIdx.x;
    o[0] = a; o[64] = b; o[128] = c; o[192] = d; o[256] = e; o[320] = f; o[384] = y - qx; o[448] = y - rx;
}
int main() {
    int h[128], *di, *dout, o[512];
    for (int i = 0; i < 64; i++) { h[i] = 1000 * i + 7; h[64 + i] = 31 * i * i + 5; }
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    const char* names[8] = {"asm v_subrev_u32_dpp quad_perm:[0,0,0,0]  (want y - qp(x))", "asm v_sub_u32_dpp    quad_perm:[0,0,0,0]  (want qp(x) - y)", "asm v_subrev_u32_dpp row_ror:14           (want y - ror(x))",
                            "asm v_sub_u32_dpp    row_ror:14           (want ror(x) - y)", "asm v_subrev_u32_dpp row_newbcast:3       (want y - bcast(x))", "asm v_subrev_u32_dpp quad_perm:[2,3,0,1]  (want y - qswap(x))",
                            "compiler: y - quad_perm[0,0,0,0](x)", "compiler: y - row_ror:14(x)"};
    for (int v = 0; v < 8; v++) {
        int right = 0, reversed = 0;
        for (int i = 0; i < 64; i++) {
            const int src = (v == 0 || v == 1 || v == 6) ? (i & ~3) : (v == 2 || v == 3 || v == 7) ? ((i & ~15) | ((i + 2) & 15)) : v == 4 ? ((i & ~15) | 3) : (i ^ 2);
            const int dp = h[src], yy = h[64 + i];
            const int want = (v == 1 || v == 3) ? dp - yy : yy - dp;
            right += o[64 * v + i] == want; reversed += o[64 * v + i] == -want;
        }
        printf("%-64s right on %2d lanes, negated on %2d\n", names[v], right, reversed);
    }
    return 0;
}
