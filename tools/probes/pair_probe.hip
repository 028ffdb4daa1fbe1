// probe: when do two wavefronts of one SIMD share a VALU issue slot on gfx950 (the 2.1-clock plateau of plain fp32 streams)?
// 512-thread workgroups = two wavefronts per SIMD; the second half of a workgroup (waves 4-7) runs ...
//   same      the same loop as the first half (same code address, started together)
//   otherop   a loop of another opcode (v_mul instead of v_add)
//   copy      a second copy of the same v_add loop at another code address
//   delayed   the same loop, entered ~1 us later
//   mixed     both halves: the same loop of 7 v_add + 1 DPP move per 8 (lock step lost at the first DPP?)
//   mixedbar  as mixed, with an s_barrier after every DPP move
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/pair_probe.hip -o tools/probes/pair_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ADD8 "v_add_f32 v0, v0, v8\n\tv_add_f32 v1, v1, v8\n\tv_add_f32 v2, v2, v8\n\tv_add_f32 v3, v3, v8\n\tv_add_f32 v4, v4, v8\n\tv_add_f32 v5, v5, v8\n\tv_add_f32 v6, v6, v8\n\tv_add_f32 v7, v7, v8\n\t"
#define MUL8 "v_mul_f32 v0, v0, v8\n\tv_mul_f32 v1, v1, v8\n\tv_mul_f32 v2, v2, v8\n\tv_mul_f32 v3, v3, v8\n\tv_mul_f32 v4, v4, v8\n\tv_mul_f32 v5, v5, v8\n\tv_mul_f32 v6, v6, v8\n\tv_mul_f32 v7, v7, v8\n\t"
#define ADD7D "v_add_f32 v0, v0, v8\n\tv_add_f32 v1, v1, v8\n\tv_add_f32 v2, v2, v8\n\tv_add_f32 v3, v3, v8\n\tv_add_f32 v4, v4, v8\n\tv_add_f32 v5, v5, v8\n\tv_add_f32 v6, v6, v8\n\tv_mov_b32_dpp v7, v7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define ADD7DB ADD7D "s_barrier\n\t"
#define ADD15D ADD8 ADD7D
#define ADD15DB ADD8 ADD7D "s_barrier\n\t"
#define X4(a) a a a a
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8"
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* clk, int iters) {
  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  if (MODE == 3 && half) __builtin_amdgcn_s_sleep(32);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0 || MODE == 3 || !half) {
    for (int i = 0; i < iters; ++i) {
      if (MODE <= 3) asm volatile(X4(ADD8) ::: CLOB);
      if (MODE == 4) asm volatile(X4(ADD7D) ::: CLOB);
      if (MODE == 5) asm volatile(X4(ADD7DB) ::: CLOB);
      if (MODE == 6) asm volatile(ADD15D ADD15D ::: CLOB);
      if (MODE == 7) asm volatile(ADD15DB ADD15DB ::: CLOB);
    }
  } else if (MODE == 1) {
    for (int i = 0; i < iters; ++i) asm volatile(X4(MUL8) ::: CLOB);
  } else if (MODE == 2) {
    for (int i = 0; i < iters; ++i) asm volatile("s_nop 0\n\t" X4(ADD8) ::: CLOB);
  } else {
    for (int i = 0; i < iters; ++i) {
      if (MODE == 4) asm volatile(X4(ADD7D) ::: CLOB);
      if (MODE == 5) asm volatile(X4(ADD7DB) ::: CLOB);
      if (MODE == 6) asm volatile(ADD15D ADD15D ::: CLOB);
      if (MODE == 7) asm volatile(ADD15DB ADD15DB ::: CLOB);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) atomicMax(clk, t1 - t0);
}
template <int MODE>
void run(const char* name, unsigned long long* clk) {
  const int iters = 4000;
  printf("%-10s:", name);
  for (int bpc : {1, 2}) {   // workgroups per CU: 2 or 4 wavefronts per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(256 * bpc), dim3(512), 0, 0, clk, 3);
    hipDeviceSynchronize();
    hipMemset(clk, 0, 8);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * bpc), dim3(512), 0, 0, clk, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("  %d waves/SIMD: %.2f clk/VALU instr/SIMD", 2 * bpc, (double)c / ((double)iters * 32 * 2 * bpc));
  }
  printf("\n");
}
int main() {
  unsigned long long* clk; hipMalloc(&clk, 8);
  run<0>("same", clk); run<1>("otherop", clk); run<2>("copy", clk); run<3>("delayed", clk);
  run<4>("mixed8", clk); run<5>("mixed8bar", clk); run<6>("mixed16", clk); run<7>("mixed16bar", clk);
  return 0;
}
