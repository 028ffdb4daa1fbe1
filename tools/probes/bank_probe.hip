// probe: does the VGPR bank (register number mod 4) of a plain VOP2 instruction's operands change its issue cost on gfx950?
// 8 independent v_mul_f32 per group with explicit registers; wall clocks from the longest-resident wavefront.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/bank_probe.hip -o tools/probes/bank_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47"
#define G8(f) f(0) f(1) f(2) f(3) f(4) f(5) f(6) f(7)
#define S(x) #x
// MODE 0: in place, second source one fixed register      v_mul v[i], v[i], v40
#define M0(i) "v_mul_f32 v" S(i) ", v" S(i) ", v40\n\t"
// MODE 1: three distinct registers, sources in different banks   v[i] = v[8+i] * v[17+i]
#define M1(i) "v_mul_f32 v" S(i) ", v[8+" S(i) "], v[17+" S(i) "]\n\t"
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* clk, int iters) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (MODE == 0) asm volatile("v_mul_f32 v0, v0, v40\n\tv_mul_f32 v1, v1, v40\n\tv_mul_f32 v2, v2, v40\n\tv_mul_f32 v3, v3, v40\n\tv_mul_f32 v4, v4, v40\n\tv_mul_f32 v5, v5, v40\n\tv_mul_f32 v6, v6, v40\n\tv_mul_f32 v7, v7, v40" ::: CLOB);
      // distinct dest, sources in different banks (8+i vs 17+i)
      if (MODE == 1) asm volatile("v_mul_f32 v0, v8, v17\n\tv_mul_f32 v1, v9, v18\n\tv_mul_f32 v2, v10, v19\n\tv_mul_f32 v3, v11, v20\n\tv_mul_f32 v4, v12, v21\n\tv_mul_f32 v5, v13, v22\n\tv_mul_f32 v6, v14, v23\n\tv_mul_f32 v7, v15, v24" ::: CLOB);
      // distinct dest, sources in the SAME bank (8+i vs 16+i)
      if (MODE == 2) asm volatile("v_mul_f32 v0, v8, v16\n\tv_mul_f32 v1, v9, v17\n\tv_mul_f32 v2, v10, v18\n\tv_mul_f32 v3, v11, v19\n\tv_mul_f32 v4, v12, v20\n\tv_mul_f32 v5, v13, v21\n\tv_mul_f32 v6, v14, v22\n\tv_mul_f32 v7, v15, v23" ::: CLOB);
      // both sources the same register
      if (MODE == 3) asm volatile("v_mul_f32 v0, v8, v8\n\tv_mul_f32 v1, v9, v9\n\tv_mul_f32 v2, v10, v10\n\tv_mul_f32 v3, v11, v11\n\tv_mul_f32 v4, v12, v12\n\tv_mul_f32 v5, v13, v13\n\tv_mul_f32 v6, v14, v14\n\tv_mul_f32 v7, v15, v15" ::: CLOB);
      // dest in the bank of a source of the NEXT instruction, all three banks equal: v[i] = v[8+i]*v[16+i], i step 4
      if (MODE == 4) asm volatile("v_mul_f32 v0, v8, v16\n\tv_mul_f32 v4, v12, v20\n\tv_mul_f32 v24, v28, v32\n\tv_mul_f32 v36, v40, v44\n\tv_mul_f32 v1, v9, v17\n\tv_mul_f32 v5, v13, v21\n\tv_mul_f32 v25, v29, v33\n\tv_mul_f32 v37, v41, v45" ::: CLOB);
      // dependent: each result feeds the next instruction two later (distance 2)
      if (MODE == 5) asm volatile("v_mul_f32 v0, v8, v17\n\tv_mul_f32 v1, v9, v18\n\tv_mul_f32 v2, v0, v19\n\tv_mul_f32 v3, v1, v20\n\tv_mul_f32 v4, v2, v21\n\tv_mul_f32 v5, v3, v22\n\tv_mul_f32 v6, v4, v23\n\tv_mul_f32 v7, v5, v24" ::: CLOB);
      // fmac: dest is also a source, two other sources in different / same banks
      if (MODE == 6) asm volatile("v_fmac_f32 v0, v8, v17\n\tv_fmac_f32 v1, v9, v18\n\tv_fmac_f32 v2, v10, v19\n\tv_fmac_f32 v3, v11, v20\n\tv_fmac_f32 v4, v12, v21\n\tv_fmac_f32 v5, v13, v22\n\tv_fmac_f32 v6, v14, v23\n\tv_fmac_f32 v7, v15, v24" ::: CLOB);
      if (MODE == 7) asm volatile("v_fmac_f32 v0, v8, v16\n\tv_fmac_f32 v1, v9, v17\n\tv_fmac_f32 v2, v10, v18\n\tv_fmac_f32 v3, v11, v19\n\tv_fmac_f32 v4, v12, v20\n\tv_fmac_f32 v5, v13, v21\n\tv_fmac_f32 v6, v14, v22\n\tv_fmac_f32 v7, v15, v23" ::: CLOB);
      // fma VOP3 with three distinct sources in three banks / one bank
      if (MODE == 8) asm volatile("v_fma_f32 v0, v8, v17, v26\n\tv_fma_f32 v1, v9, v18, v27\n\tv_fma_f32 v2, v10, v19, v28\n\tv_fma_f32 v3, v11, v20, v29\n\tv_fma_f32 v4, v12, v21, v30\n\tv_fma_f32 v5, v13, v22, v31\n\tv_fma_f32 v6, v14, v23, v32\n\tv_fma_f32 v7, v15, v24, v33" ::: CLOB);
      if (MODE == 9) asm volatile("v_fma_f32 v0, v8, v16, v24\n\tv_fma_f32 v1, v9, v17, v25\n\tv_fma_f32 v2, v10, v18, v26\n\tv_fma_f32 v3, v11, v19, v27\n\tv_fma_f32 v4, v12, v20, v28\n\tv_fma_f32 v5, v13, v21, v29\n\tv_fma_f32 v6, v14, v22, v30\n\tv_fma_f32 v7, v15, v23, v31" ::: CLOB);
      // v_mov_b32 (VOP1)
      if (MODE == 10) asm volatile("v_mov_b32 v0, v8\n\tv_mov_b32 v1, v9\n\tv_mov_b32 v2, v10\n\tv_mov_b32 v3, v11\n\tv_mov_b32 v4, v12\n\tv_mov_b32 v5, v13\n\tv_mov_b32 v6, v14\n\tv_mov_b32 v7, v15" ::: CLOB);
      // cndmask_e32 with vcc written in the kernel prologue
      if (MODE == 11) asm volatile("v_cndmask_b32_e32 v0, v8, v17, vcc\n\tv_cndmask_b32_e32 v1, v9, v18, vcc\n\tv_cndmask_b32_e32 v2, v10, v19, vcc\n\tv_cndmask_b32_e32 v3, v11, v20, vcc\n\tv_cndmask_b32_e32 v4, v12, v21, vcc\n\tv_cndmask_b32_e32 v5, v13, v22, vcc\n\tv_cndmask_b32_e32 v6, v14, v23, vcc\n\tv_cndmask_b32_e32 v7, v15, v24, vcc" ::: CLOB);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) atomicMax(clk, t1 - t0);
}
template <int MODE>
void run(const char* name, unsigned long long* clk) {
  const int iters = 2000;
  printf("%-44s:", name);
  for (int wps : {1, 2, 4, 8}) {
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, clk, 3);
    hipDeviceSynchronize();
    hipMemset(clk, 0, 8);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), 0, 0, clk, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    printf("  w%d: %.2f", wps, (double)c / ((double)iters * 64 * wps));
  }
  printf("   clk/instr/SIMD\n");
}
int main() {
  unsigned long long* clk; hipMalloc(&clk, 8);
  run<0>("v_mul in place, fixed 2nd source", clk);
  run<1>("v_mul 3 distinct regs, sources other banks", clk);
  run<2>("v_mul 3 distinct regs, sources SAME bank", clk);
  run<3>("v_mul both sources one register", clk);
  run<4>("v_mul dest+sources all one bank", clk);
  run<5>("v_mul dependent at distance 2", clk);
  run<6>("v_fmac sources other banks", clk);
  run<7>("v_fmac sources same bank", clk);
  run<8>("v_fma 3 sources, 3 banks", clk);
  run<9>("v_fma 3 sources, 1 bank", clk);
  run<10>("v_mov", clk);
  run<11>("v_cndmask_e32 vcc", clk);
  return 0;
}
