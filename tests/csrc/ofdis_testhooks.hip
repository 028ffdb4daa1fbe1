// ofdis_testhooks.hip -- TEST LIBRARY (libofdis_testhooks.so), not part of the product: device kernels that expose the
// header-only arithmetic helpers of of_dis_amd/csrc/ofdis_dev.h (wave reduction order, trimmed divide / square root) and
// the host-side outlier threshold to the parity tests.  The shipped libofdis_hip.so exports include/ofdis.h and nothing else.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../of_dis_amd/csrc/ofdis_dev.h"

using namespace ofdis;
using namespace ofdis::exact;  // (the helpers under test are the exact contract's)

__global__ void wave_sum_test_kernel(const float* in, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = wave_sum(in[i]);
}
// out[6][n]: div_rn(a,b), a/b, sqrt_rn(|a|), sqrtf(|a|), the fused TV kernel's quotient a/b (shared refined reciprocal, no
// v_div_fixup) and its quotient by a root b / sqrt(|a|)
__global__ void div_sqrt_test_kernel(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i], y = b[i];
  out[i] = div_rn(x, y);
  out[n + i] = x / y;
  out[2 * n + i] = sqrt_rn(fabsf(x));
  out[3 * n + i] = sqrtf(fabsf(x));
  out[4 * n + i] = div_by_finite(x, 0.0f - y, rcp_refined(y));
  const float sq = sqrt_rn(fabsf(x));
  out[5 * n + i] = div_by_finite(y, 0.0f - sq, rcp_refined(sq));
}

extern "C" {
int ofdis_test_wave_sum(const float* in, float* out, int n, void* stream) {
  hipLaunchKernelGGL(wave_sum_test_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, out, n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
int ofdis_test_div_sqrt(const float* a, const float* b, float* out, int n, void* stream) {
  hipLaunchKernelGGL(div_sqrt_test_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
// host only, needs no GPU: the squared outlier threshold of the patch kernels
float ofdis_test_outlier_sq(float t) { return outlier_sq_threshold(t); }
}
