// Micro-benchmarks that size the integer-VALU roofline for the BLS12-381 kernels on gfx950.
// Measures: v_mad_u64_u32, v_mul_lo/hi_u32, 24-bit mads, 64-bit add, f64 fma throughput,
// and the cost of streaming straight-line code larger than the instruction cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

typedef unsigned long long u64; typedef unsigned int u32;

template<int MODE> __global__ void __launch_bounds__(256) k_alu(u32* out, int iters, u32 seed) {
  u32 a = seed + threadIdx.x, b = seed*3 + blockIdx.x;
  u64 acc[8]; u32 r[8]; double d[8];
  #pragma unroll
  for (int i=0;i<8;i++){acc[i]=a+i; r[i]=a*7+i; d[i]=1.0+i+a;}
  for (int it=0; it<iters; ++it) {
    #pragma unroll
    for (int rep=0; rep<4; ++rep) {
      #pragma unroll
      for (int i=0;i<8;i++) {
        if (MODE==0) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc"); }
        else if (MODE==1) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b)); }
        else if (MODE==2) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r[i]) : "v"(b)); }
        else if (MODE==3) { asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(r[i]) : "v"(b)); }
        else if (MODE==4) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(b)); }
        else if (MODE==5) { asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(b) : "vcc"); }
        else if (MODE==6) { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i+1)&7])); }
        else if (MODE==7) { asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[i])); }
        else if (MODE==8) { asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(r[i]) : "v"(b) : "vcc"); }
        else if (MODE==9) { asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(b)); }
        else if (MODE==10) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(b)); }
        else if (MODE==11) { asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(r[i]) :: "a0"); }
      }
    }
  }
  u32 s=0;
  #pragma unroll
  for (int i=0;i<8;i++) s += (u32)acc[i] + (u32)(acc[i]>>32) + r[i] + (u32)d[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))
#define R256(x) R4(R64(x))
#define R1024(x) R4(R256(x))
#define R4096(x) R4(R1024(x))
// straight-line body of N v_add_u32 (4 B each... with literal 8 B). body bytes = N*8.
template<int KB> __global__ void __launch_bounds__(256) k_icache(u32* out, int iters, u32 seed) {
  u32 r0 = seed + threadIdx.x, r1 = seed ^ blockIdx.x, r2 = 3, r3 = 4;
  for (int it=0; it<iters; ++it) {
    if (KB==8)   { R256(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x54321, %1\n v_add_u32 %2, 0x11111, %2\n v_add_u32 %3, 0x22222, %3" : "+v"(r0),"+v"(r1),"+v"(r2),"+v"(r3));) }
    if (KB==32)  { R1024(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x54321, %1\n v_add_u32 %2, 0x11111, %2\n v_add_u32 %3, 0x22222, %3" : "+v"(r0),"+v"(r1),"+v"(r2),"+v"(r3));) }
    if (KB==128) { R4096(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x54321, %1\n v_add_u32 %2, 0x11111, %2\n v_add_u32 %3, 0x22222, %3" : "+v"(r0),"+v"(r1),"+v"(r2),"+v"(r3));) }
    if (KB==512) { R4(R4096(asm volatile("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x54321, %1\n v_add_u32 %2, 0x11111, %2\n v_add_u32 %3, 0x22222, %3" : "+v"(r0),"+v"(r1),"+v"(r2),"+v"(r3));)) }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=r0+r1+r2+r3;
}

template<typename F> float timeit(F f, int reps=3) {
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best=1e30f;
  for(int i=0;i<reps;i++){ hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms,e0,e1); if(ms<best)best=ms; }
  return best;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int ncu = p.multiProcessorCount;
  u32* out; CK(hipMalloc(&out, sizeof(u32)*ncu*64*256));
  const char* names[]={"v_mad_u64_u32","v_mul_lo_u32","v_mul_hi_u32","v_mad_u32_u24","v_mul_hi_u32_u24","v_add_co_u32","v_fma_f64","v_lshrrev_b64","v_addc_co_u32","v_add3_u32","v_cndmask_b32","accvgpr_wr+rd"};
  for (int wpc : {4, 8, 16}) {   // waves per CU (blocks of 256 threads = 4 waves)
    int blocks = ncu * (wpc/4);
    for (int mode=0; mode<12; ++mode) {
      int iters = 20000;
      auto launch=[&](){
        switch(mode){
          case 0: hipLaunchKernelGGL(k_alu<0>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 1: hipLaunchKernelGGL(k_alu<1>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 2: hipLaunchKernelGGL(k_alu<2>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 3: hipLaunchKernelGGL(k_alu<3>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 4: hipLaunchKernelGGL(k_alu<4>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 5: hipLaunchKernelGGL(k_alu<5>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 6: hipLaunchKernelGGL(k_alu<6>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 7: hipLaunchKernelGGL(k_alu<7>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 8: hipLaunchKernelGGL(k_alu<8>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 9: hipLaunchKernelGGL(k_alu<9>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 10: hipLaunchKernelGGL(k_alu<10>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
          case 11: hipLaunchKernelGGL(k_alu<11>,dim3(blocks),dim3(256),0,0,out,iters,1u);break;
        }};
      float ms = timeit(launch);
      double ops = (double)blocks*256*iters*32;   // lane-ops
      double waveinstr_per_simd = (double)iters*32*(wpc/4);  // wave-instructions per SIMD
      printf("ALU wpc=%2d %-18s %8.3f ms  %8.2f Tlane-op/s  %6.2f ns/wave-instr/SIMD (=%.2f cyc @2.4GHz)\n", wpc, names[mode], ms, ops/ms/1e9, ms*1e6/waveinstr_per_simd, ms*1e6/waveinstr_per_simd*2.4);
    }
  }
  // I-cache streaming: blocks = ncu (1 wave/SIMD), same total instruction count
  {
    int blocks=ncu;
    struct {int kb; int iters;} cfg[]={{8,4096},{32,1024},{128,256},{512,64}};
    for (auto c: cfg) {
      auto launch=[&](){
        if(c.kb==8) hipLaunchKernelGGL(k_icache<8>,dim3(blocks),dim3(256),0,0,out,c.iters,1u);
        if(c.kb==32) hipLaunchKernelGGL(k_icache<32>,dim3(blocks),dim3(256),0,0,out,c.iters,1u);
        if(c.kb==128) hipLaunchKernelGGL(k_icache<128>,dim3(blocks),dim3(256),0,0,out,c.iters,1u);
        if(c.kb==512) hipLaunchKernelGGL(k_icache<512>,dim3(blocks),dim3(256),0,0,out,c.iters,1u);
      };
      float ms=timeit(launch);
      double instr=(double)c.iters*c.kb*1024/8;
      printf("ICACHE body=%3d KB iters=%4d  %8.3f ms  %.2f ns/instr/wave (=%.2f cyc)\n", c.kb, c.iters, ms, ms*1e6/instr, ms*1e6/instr*2.4);
    }
    // same with 1 block only per 2 CUs? and with 64-thread blocks (1 wave per CU)
  }
  return 0;
}
