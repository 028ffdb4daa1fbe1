// probe: does a long straight-line VALU loop (larger than the instruction buffers, like the fused TV kernel's 12 KB body)
// run slower per instruction than a short one?  Same instructions, loop bodies of 64 ... 8192 instructions.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_probe.hip -o tools/probes/fetch_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int BODY, int ENC8>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float x[8], y[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { x[c] = seed + c + threadIdx.x; y[c] = seed * 0.5f + c; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < BODY / 8; ++r) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (ENC8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(y[c]), "v"(seed));   // 8-byte encoding
        else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(y[c]));                        // 4-byte encoding
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += x[c] + y[c];
  if (s == 12345.678f) out[0] = s;
}

template <int BODY, int ENC8>
void run(float* out) {
  const long total = 1 << 22;  // instructions per wave
  const int iters = (int)(total / BODY);
  printf("%s body=%5d instr (%6d B):", ENC8 ? "v_fma(8B)" : "v_mul(4B)", BODY, BODY * (ENC8 ? 8 : 4));
  for (int wps : {1, 2, 3, 4}) {
    const int blocks = 256 * wps;  // 4 waves per block, wps blocks per CU
    k<BODY, ENC8><<<blocks, 256>>>(out, 4, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<BODY, ENC8><<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  w%d: %.2f ns/instr/SIMD", wps, ms * 1e6 / ((double)iters * BODY * wps));
  }
  printf("\n");
}

int main() {
  float* out; hipMalloc(&out, 4);
  run<64, 0>(out); run<512, 0>(out); run<2048, 0>(out); run<4096, 0>(out); run<8192, 0>(out); run<16384, 0>(out);
  run<64, 1>(out); run<512, 1>(out); run<2048, 1>(out); run<4096, 1>(out); run<8192, 1>(out);
  return 0;
}
