// HBM counter calibration: a copy of N doubles with the step kernel's access shape (8 B per lane, one wave reading
// and writing a contiguous run).  Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE to see what the counters report
// for a known byte count (MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for 16 B/lane streams).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_copy8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1.0;
}
int main() {
  const size_t n = 64ull << 20;   // 64 Mi doubles = 512 MiB each way (beyond L2 and Infinity Cache)
  double *a, *b;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
  hipMemset(a, 0, n * 8);
  for (int it = 0; it < 3; it++) hipLaunchKernelGGL(calib_copy8, dim3((n + 63) / 64), dim3(64), 0, 0, a, b, n);
  hipDeviceSynchronize();
  printf("calib_copy8: %zu bytes read and %zu bytes written per launch\n", n * 8, n * 8);
  return 0;
}
