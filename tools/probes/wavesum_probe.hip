// diagnostic: dump every stage of the butterfly for x[lane] = lane
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x, float old = 0.0f) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float x = (float)l;
  out[0 * 64 + l] = dpp_mov<0xB1>(x);
  out[1 * 64 + l] = dpp_mov<0x4E>(x);
  out[2 * 64 + l] = dpp_mov<0x141>(x);
  out[3 * 64 + l] = dpp_mov<0x140>(x);
  auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  out[4 * 64 + l] = __builtin_bit_cast(float, r[0]);
  out[5 * 64 + l] = __builtin_bit_cast(float, r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  out[6 * 64 + l] = __builtin_bit_cast(float, q[0]);
  out[7 * 64 + l] = __builtin_bit_cast(float, q[1]);
  out[8 * 64 + l] = dpp_mov<0x138>(x, -1.0f);
  out[9 * 64 + l] = dpp_mov<0x130>(x, -1.0f);
  float y = x + dpp_mov<0xB1>(x);
  out[10 * 64 + l] = y;
}
int main() {
  float* d; hipMalloc(&d, 11 * 64 * 4);
  k<<<1, 64>>>(d);
  float h[11 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"quad1032", "quad2301", "half_mirror", "row_mirror", "pl16[0]", "pl16[1]", "pl32[0]", "pl32[1]", "wave_shr", "wave_shl", "x+quad1032"};
  for (int s = 0; s < 11; ++s) { printf("%-12s", names[s]); for (int l = 0; l < 64; ++l) printf(" %g", h[s * 64 + l]); printf("\n"); }
  return 0;
}
