// Semantics check of v_permlane32_swap on gfx950 for the wide register solver: with both operands holding v,
// result[0] should carry v's lanes 0..31 in both halves, result[1] v's lanes 32..63 in both halves.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int v = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  out[threadIdx.x] = r[0];
  out[64 + threadIdx.x] = r[1];
}
int main() {
  int* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r[0]:"); for (int i = 0; i < 64; i++) printf(" %d", h[i]); printf("\n");
  printf("r[1]:"); for (int i = 0; i < 64; i++) printf(" %d", h[64 + i]); printf("\n");
  return 0;
}
