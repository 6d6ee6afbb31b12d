// Micro-benchmark 2: (a) does rotating the dead carry-out SGPR of v_mad_u64_u32 remove the
// 1-wave/SIMD penalty?  (b) cycles per 14x28-bit signed-limb Montgomery multiplication (the device
// field representation) at 1/2/4 waves per SIMD.  (c) v_cndmask / v_bfi / DPP mov costs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64; typedef unsigned int u32; typedef int i32; typedef long long i64;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template<int MODE> __global__ void __launch_bounds__(256) k_alu(u32* out, int iters, u32 seed) {
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x;
  u64 acc[8]; u32 r[8];
  #pragma unroll
  for (int i=0;i<8;i++){acc[i]=a+i; r[i]=a*7+i;}
  for (int it=0; it<iters; ++it) {
    #pragma unroll
    for (int rep=0; rep<4; ++rep) {
      if (MODE==0) { // rotating sdst
        asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[22:23], %8, %9, %1\n v_mad_u64_u32 %2, s[24:25], %8, %9, %2\n v_mad_u64_u32 %3, s[26:27], %8, %9, %3\n"
                     "v_mad_u64_u32 %4, s[28:29], %8, %9, %4\n v_mad_u64_u32 %5, s[30:31], %8, %9, %5\n v_mad_u64_u32 %6, s[32:33], %8, %9, %6\n v_mad_u64_u32 %7, s[34:35], %8, %9, %7"
          : "+v"(acc[0]),"+v"(acc[1]),"+v"(acc[2]),"+v"(acc[3]),"+v"(acc[4]),"+v"(acc[5]),"+v"(acc[6]),"+v"(acc[7]) : "v"(a),"v"(b)
          : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
      } else if (MODE==1) { // same sdst (vcc)
        #pragma unroll
        for (int i=0;i<8;i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      } else if (MODE==2) { // signed mad
        #pragma unroll
        for (int i=0;i<8;i++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc");
      } else if (MODE==3) { // mad interleaved 1:1 with plain valu
        #pragma unroll
        for (int i=0;i<8;i++) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_add_u32 %1, %1, %3" : "+v"(acc[i]), "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
      } else if (MODE==4) { // cndmask with sgpr pair cond (VOP3)
        #pragma unroll
        for (int i=0;i<8;i++) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(b) : "s20","s21");
      } else if (MODE==5) { // bfi
        #pragma unroll
        for (int i=0;i<8;i++) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      } else if (MODE==6) { // dpp mov (quad_perm swap neighbours)
        #pragma unroll
        for (int i=0;i<8;i++) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
      } else if (MODE==7) { // ashr i64
        #pragma unroll
        for (int i=0;i<8;i++) asm volatile("v_ashrrev_i64 %0, 28, %0" : "+v"(acc[i]));
      } else if (MODE==8) { // cndmask vop2 with vcc set once
        #pragma unroll
        for (int i=0;i<8;i++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b));
      }
    }
  }
  u32 s=0;
  #pragma unroll
  for (int i=0;i<8;i++) s += (u32)acc[i] + (u32)(acc[i]>>32) + r[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}

// 14 x 28-bit signed limbs, Montgomery R = 2^392, product-scanning (FIPS), no carries between lanes/words.
#define NL 15
#define MASK28 0x07ffffff
#define LBITS 27
__constant__ i32 QL[NL] = {0x7ffaaab,0x7fdffff,0x7fffdcf,0x7d62a7f,0x241eabf,0x7b0f624,0x349a83d,0x4afd9cc,0x4e9c4e1,0x35d91dd,0x5a10d2e,0x4d258dd,0x65cbfe6,0x47a8e5f,0x6a04};
struct F { i32 v[NL]; };
__device__ __forceinline__ F fmul(const F& a, const F& b, i32 qinv) {
  i32 m[NL]; F r; i64 acc=0;
  #pragma unroll
  for (int k=0;k<NL;k++) {
    #pragma unroll
    for (int i=0;i<=k;i++) acc += (i64)a.v[i]*b.v[k-i];
    #pragma unroll
    for (int i=0;i<k;i++) acc += (i64)m[i]*QL[k-i];
    m[k] = ((i32)acc*qinv) & MASK28;
    acc += (i64)m[k]*QL[0];
    acc >>= LBITS;
  }
  #pragma unroll
  for (int k=NL;k<2*NL-1;k++) {
    #pragma unroll
    for (int i=k-NL+1;i<NL;i++) acc += (i64)a.v[i]*b.v[k-i];
    #pragma unroll
    for (int i=k-NL+1;i<NL;i++) acc += (i64)m[i]*QL[k-i];
    r.v[k-NL] = (i32)acc & MASK28;
    acc >>= LBITS;
  }
  r.v[NL-1]=(i32)acc;
  return r;
}
template<int WPS> __global__ void __launch_bounds__(256, WPS) k_fmul(i32* io, int iters, i32 qinv) {
  F a,b; int t=blockIdx.x*blockDim.x+threadIdx.x;
  #pragma unroll
  for(int i=0;i<NL;i++){a.v[i]=(io[i]+t)&MASK28; b.v[i]=(io[NL+i]^t)&MASK28;}
  for(int it=0;it<iters;it++){ F c=fmul(a,b,qinv); a=b; b=c; }
  i32 s=0;
  #pragma unroll
  for(int i=0;i<NL;i++) s^=b.v[i];
  io[64+t%64]=s;
}

template<typename Fn> float timeit(Fn f, int reps=3) {
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  float best=1e30f;
  for(int i=0;i<reps;i++){ (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms,e0,e1); if(ms<best)best=ms; }
  return best;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0)); int ncu=p.multiProcessorCount;
  u32* out; CK(hipMalloc(&out, sizeof(u32)*ncu*64*256)); CK(hipMemset(out,1,4096));
  const char* names[]={"mad_u64 rotating sdst","mad_u64 sdst=vcc","mad_i64_i32 sdst=vcc","mad_u64+v_add pair","cndmask_e64 sgpr","v_bfi_b32","v_mov_dpp quad swap","v_ashrrev_i64","cndmask vop2 vcc"};
  for (int wpc : {4, 8, 16}) {
    int blocks = ncu*(wpc/4), iters=20000;
    for (int mode=0; mode<9; ++mode) {
      auto launch=[&](){ switch(mode){
        case 0: hipLaunchKernelGGL(k_alu<0>,dim3(blocks),dim3(256),0,0,out,iters,1u);break; case 1: hipLaunchKernelGGL(k_alu<1>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
        case 2: hipLaunchKernelGGL(k_alu<2>,dim3(blocks),dim3(256),0,0,out,iters,1u);break; case 3: hipLaunchKernelGGL(k_alu<3>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
        case 4: hipLaunchKernelGGL(k_alu<4>,dim3(blocks),dim3(256),0,0,out,iters,1u);break; case 5: hipLaunchKernelGGL(k_alu<5>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
        case 6: hipLaunchKernelGGL(k_alu<6>,dim3(blocks),dim3(256),0,0,out,iters,1u);break; case 7: hipLaunchKernelGGL(k_alu<7>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
        case 8: hipLaunchKernelGGL(k_alu<8>,dim3(blocks),dim3(256),0,0,out,iters,1u);break; }};
      float ms=timeit(launch); double n=(double)iters*32*(wpc/4)*(mode==3?2:1);
      printf("ALU wpc=%2d %-24s %8.3f ms  %6.2f cyc/wave-instr/SIMD @2.4GHz\n", wpc, names[mode], ms, ms*1e6/n*2.4);
    }
  }
  int iters=2000;
  for (int wps : {1,2,4}) {
    int blocks=ncu*wps;
    auto launch=[&](){ if(wps==1) hipLaunchKernelGGL(k_fmul<1>,dim3(blocks),dim3(256),0,0,(i32*)out,iters,(i32)0xfffcfffd);
                       if(wps==2) hipLaunchKernelGGL(k_fmul<2>,dim3(blocks),dim3(256),0,0,(i32*)out,iters,(i32)0xfffcfffd);
                       if(wps==4) hipLaunchKernelGGL(k_fmul<4>,dim3(blocks),dim3(256),0,0,(i32*)out,iters,(i32)0xfffcfffd); };
    float ms=timeit(launch);
    double muls=(double)blocks*256*iters;
    printf("FMUL15x27 waves/SIMD=%d  %8.3f ms  %.1f Gmul/s  %.0f cyc/mul/wave-slot (@2.4GHz, per SIMD: %.0f)\n", wps, ms, muls/ms/1e6, ms*1e6*2.4/iters, ms*1e6*2.4/iters/wps);
  }
  return 0;
}
