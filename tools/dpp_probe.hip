// dpp_probe.hip -- which lane does a lane read from under the DPP controls the lane-row layout (row_body.inc) uses?  Prints, for the
// first row of 16 lanes, the source lane of row_ror:n, of a bank-masked write, of row_newbcast:n and of the quad permutes.
//   hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o tools/dpp_probe && tools/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK> __device__ int dpp(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, BANK, BANK == 0xf); }
__global__ void k(int* out) {
    const int x = threadIdx.x;
    int r = 0;
    out[0 * 64 + x] = dpp<0x120 + 2, 0xf>(0, x);        // row_ror:2
    out[1 * 64 + x] = dpp<0x120 + 12, 0xf>(0, x);       // row_ror:12
    out[2 * 64 + x] = dpp<0x120 + 8, 0x4>(100 + x, x);  // row_ror:8 into bank 2 only (old = 100 + lane elsewhere)
    out[3 * 64 + x] = dpp<0x150 + 5, 0xf>(0, x);        // row_newbcast:5
    out[4 * 64 + x] = dpp<0x4E, 0xf>(0, x);             // quad_perm [2,3,0,1]
    out[5 * 64 + x] = dpp<0xB1, 0xf>(0, x);             // quad_perm [1,0,3,2]
    out[6 * 64 + x] = dpp<0x44, 0xf>(0, x);             // quad_perm [0,1,0,1]
    out[7 * 64 + x] = dpp<0xEE, 0xf>(0, x);             // quad_perm [2,3,2,3]
    (void)r;
}
int main() {
    int* d; hipMalloc(&d, 8 * 64 * 4);
    k<<<1, 64>>>(d);
    int h[8 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* name[8] = {"row_ror:2", "row_ror:12", "row_ror:8 bank 0x4", "row_newbcast:5", "quad_perm[2,3,0,1]", "quad_perm[1,0,3,2]", "quad_perm[0,1,0,1]", "quad_perm[2,3,2,3]"};
    for (int j = 0; j < 8; j++) { printf("%-22s", name[j]); for (int i = 0; i < 32; i++) printf(" %3d", h[j * 64 + i]); printf("\n"); }
    return 0;
}
