// probe: achievable HBM bandwidth of streaming kernels at the byte counts of the warp kernel
// (read 2 planes + write 2 planes of N floats), for several N.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k22(const f4* a, const f4* b, f4* c, f4* d, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f4 x = a[i], y = b[i];
    c[i] = x + y;
    d[i] = x - y;
  }
}
__global__ __launch_bounds__(256) void k11(const f4* a, f4* c, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) c[i] = a[i];
}
int main() {
  const long long nmax = 64ll << 20;  // floats per plane
  float *a, *b, *c, *d;
  hipMalloc(&a, nmax * 4); hipMalloc(&b, nmax * 4); hipMalloc(&c, nmax * 4); hipMalloc(&d, nmax * 4);
  hipMemset(a, 0, nmax * 4); hipMemset(b, 0, nmax * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (long long n : {3670016ll, 14680064ll, 29360128ll, 58720256ll}) {
    for (int grid : {2048, 8192, 0}) {
      const long long n4 = n / 4;
      int g = grid ? grid : (int)((n4 + 255) / 256);
      for (int w = 0; w < 3; ++w) k22<<<g, 256>>>((f4*)a, (f4*)b, (f4*)c, (f4*)d, n4);
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) k22<<<g, 256>>>((f4*)a, (f4*)b, (f4*)c, (f4*)d, n4);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("2r2w n=%lld grid=%d: %.1f us/launch, %.0f GB/s\n", n, g, ms / reps * 1e3, 16.0 * n / (ms / reps * 1e-3) / 1e9);
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) k11<<<g, 256>>>((f4*)a, (f4*)c, n4);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("1r1w n=%lld grid=%d: %.1f us/launch, %.0f GB/s\n", n, g, ms / reps * 1e3, 8.0 * n / (ms / reps * 1e-3) / 1e9);
    }
  }
  return 0;
}
