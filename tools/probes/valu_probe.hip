// probe: VALU issue rate on gfx950 as a function of waves per SIMD and independent chains per wave, for
// plain v_add_f32, v_pk_add_f32, DPP adds, IEEE division and v_cndmask.  Prints cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2v __attribute__((ext_vector_type(2)));

template <int CH, int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  float x[CH];
  unsigned long long m64 = __builtin_amdgcn_read_exec() ^ (unsigned long long)iters;
  asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(seed), "v"(1.5f) : "vcc");
  f2v p[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { x[c] = seed + c + threadIdx.x; p[c] = f2v{x[c], x[c] + 1}; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (MODE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[c]) : "v"(p[(c + 1) % CH]));
        if (MODE == 2) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[c]));
        if (MODE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[c]));
        if (MODE == 6) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[c]));
        if (MODE == 7) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" : "+v"(x[c]));
        if (MODE == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(seed));
        if (MODE == 9) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 10) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 11) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(seed), "s"(m64));
        if (MODE == 12) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(x[c]), "v"(seed) : "vcc");
        if (MODE == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 14) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 15) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 16) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[c]) : "v"(seed) : "vcc");
        if (MODE == 17) asm volatile("v_cmp_gt_f32_e64 %2, %0, %1" : : "v"(x[c]), "v"(seed), "s"(m64));
        if (MODE == 18) asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 19) asm volatile("v_add_f32 %0, %0, |%1|" : "+v"(x[c]) : "v"(seed));
        if (MODE == 20) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[c]) : "v"(seed));
        if (MODE == 21) asm volatile("v_add_f32 %0, %0, %1\n s_nop 1" : "+v"(x[c]) : "v"(seed));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c] + p[c].x + p[c].y;
  if (s == 12345.678f) out[0] = s;
}

template <int CH, int MODE>
void run(const char* name, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 4, 8}) {                  // waves per SIMD
    const int threads = 256, blocks = 256 * wps;  // 4 waves per block -> one per SIMD; wps blocks per CU
    k<CH, MODE><<<blocks, threads>>>(out, 10, 1.0f);
    hipEventRecord(e0);
    k<CH, MODE><<<blocks, threads>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 16 * CH * wps;
    printf("%-12s chains=%d waves/SIMD=%d : %.2f cycles per wave-instr per SIMD (2.4 GHz)\n", name, CH, wps,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
  }
}

int main() {
  float* out; hipMalloc(&out, 4);
  run<4, 0>("v_add", out);
  run<4, 8>("cndmask_vcc", out); run<4, 11>("cndmask_sgpr", out); run<4, 16>("cmp+nop+cnd", out);
  run<4, 10>("v_bfi", out); run<4, 12>("v_cmp_vcc", out); run<4, 17>("v_cmp_sgpr", out);
  run<4, 13>("v_max", out); run<4, 14>("v_add_u32", out); run<4, 15>("v_and", out); run<4, 18>("v_mov", out);
  run<4, 19>("v_add_abs", out); run<4, 20>("v_lshl_add", out); run<4, 21>("v_add+nop1", out);
  return 0;
}
