// How many single-wave workgroups with B bytes of dynamic LDS does a CU of this chip really hold?  Each block spins
// for a fixed time; 256 CUs x K blocks finish in one spin time if K fit per CU, in two otherwise.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void spin(long long ticks, int* out) {
  extern __shared__ double lds[];
  long long t0 = clock64();
  lds[threadIdx.x] = (double)t0;
  while (clock64() - t0 < ticks) { }
  if (lds[threadIdx.x] == 1.5) out[0] = 1;
}
int main() {
  int* out; hipMalloc(&out, 4);
  hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  int sizes[] = {20480, 21504, 22528, 23040, 23296, 23352, 23400, 23552, 24576, 26624, 27136, 27304};
  for (int k : {6, 7, 8}) {
    for (int b : sizes) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(spin, dim3(256 * k), dim3(64), b, 0, 200000LL, out);      // warm
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(spin, dim3(256 * k), dim3(64), b, 0, 200000LL, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("blocks per CU wanted %d, LDS %6d B: %.3f ms\n", k, b, ms);
    }
  }
  return 0;
}
