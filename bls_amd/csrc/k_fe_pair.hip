// k_fe_pair.hip -- the final-exponentiation kernels of the lane-pair layout (pair_kernels.inc: k_final_exp_pair, k_final_exp_is_one_pair),
// a translation unit of their own because they are compiled with the multiply cores as ASSEMBLY BLOBS behind inline-asm statements
// (BLSMI_ASM_CORES: core_asm.inc, gen_core_asm.py) instead of out-of-line functions: every compiled function begins with
// s_waitcnt vmcnt(0), which drains the caller's scratch stores at each of the ~5 000 products of a final exponentiation; an asm statement
// does not.  Same-box A/B: k_final_exp_pair 11.39 -> 10.93 ms.  The Miller-loop kernels keep the function cores (9.46 -> 9.53 ms with blobs).
#define BLSMI_ASM_CORES
#include "pairing.cuh"
#include "device_io.cuh"

#define BLSMI_PAIR_FE_ONLY
#include "pair_kernels.inc"
