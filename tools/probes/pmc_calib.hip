// probe: calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 against known byte counts, for the access
// widths this repository uses (4 B/lane and 16 B/lane), working set 940 MB (beyond the 256 MB Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void calib_copy_b32(const float* a, float* c, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) c[i] = a[i];
}
__global__ __launch_bounds__(256) void calib_copy_b128(const f4* a, f4* c, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) c[i] = a[i];
}
int main() {
  const long long n = 117440512ll;  // floats: 470 MB read + 470 MB written per launch
  float *a, *c;
  hipMalloc(&a, n * 4); hipMalloc(&c, n * 4);
  hipMemset(a, 0, n * 4);
  for (int r = 0; r < 3; ++r) {
    calib_copy_b32<<<8192, 256>>>(a, c, n);
    calib_copy_b128<<<8192, 256>>>((const f4*)a, (f4*)c, n / 4);
  }
  hipDeviceSynchronize();
  printf("bytes read = bytes written = %lld per launch\n", n * 4);
  return 0;
}
