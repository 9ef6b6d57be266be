// What slows a wave of the step kernel down when its CU (and the chip) fills up?  One 64-lane workgroup with a
// 39 KiB LDS image per block, like the step kernel; per-wave cycles of four micro-workloads, averaged over all waves,
// for launches of 64 / 256 / 1024 / 4096 blocks:
//   fma_loop   512 dependent fp64 mul+add in a loop                      (VALU latency, code resident)
//   fma_flat   the same chain, fully unrolled: ~8 KiB of straight-line code per copy, REP distinct copies
//              (instruction fetch: the step kernel is ~130 KiB of mostly straight-line code)
//   lds        256 write -> other-lane-read round trips
//   table      128 dependent reads of a table every wave shares (first touch in this launch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int K>
__device__ __forceinline__ double flat_chain(double x, double y) {
#pragma unroll
  for (int i = 0; i < 256; i++) x = x * y + (0.5 + K);
  return x;
}

__global__ __launch_bounds__(64) void probe(double* out, const int* idx, unsigned long long* cyc) {
  extern __shared__ double lds[];
  int L = threadIdx.x;
  double x = out[L] + 1.0, y = 1.000001;
  long long t0, t1;
  t0 = clock64();
  for (int i = 0; i < 512; i++) x = x * y + 0.5;
  t1 = clock64(); if (L == 0) atomicAdd(cyc + 0, (unsigned long long)(t1 - t0));
  asm volatile("" : "+v"(x));
  t0 = clock64();
  asm volatile("" : "+v"(x));
  x = flat_chain<0>(x, y); x = flat_chain<1>(x, y); x = flat_chain<2>(x, y); x = flat_chain<3>(x, y);
  x = flat_chain<4>(x, y); x = flat_chain<5>(x, y); x = flat_chain<6>(x, y); x = flat_chain<7>(x, y);
  x = flat_chain<8>(x, y); x = flat_chain<9>(x, y); x = flat_chain<10>(x, y); x = flat_chain<11>(x, y);
  x = flat_chain<12>(x, y); x = flat_chain<13>(x, y); x = flat_chain<14>(x, y); x = flat_chain<15>(x, y);
  asm volatile("" : "+v"(x));
  t1 = clock64();
  asm volatile("" : "+v"(x)); if (L == 0) atomicAdd(cyc + 1, (unsigned long long)(t1 - t0));
  lds[L] = x;
  t0 = clock64();
  for (int i = 0; i < 256; i++) {
    lds[L + 64 * (i & 31)] = x; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    x = lds[((L + 1) & 63) + 64 * (i & 31)] + 1.0; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  t1 = clock64(); if (L == 0) atomicAdd(cyc + 2, (unsigned long long)(t1 - t0));
  int j = L;
  t0 = clock64();
  for (int i = 0; i < 128; i++) j = idx[j];
  t1 = clock64(); if (L == 0) atomicAdd(cyc + 3, (unsigned long long)(t1 - t0));
  if (x == 12345.678 && j == -1) out[L] = x + j;
}

int main() {
  double* out; int* idx; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMemset(out, 0, 64 * 8);
  const int T = 8192;       // 32 KiB table, a permutation with one long cycle so that every read is a new line
  int* h = (int*)malloc(T * 4);
  for (int i = 0; i < T; i++) h[i] = (i + 64 * 37 + 16) % T;
  hipMalloc(&idx, T * 4); hipMemcpy(idx, h, T * 4, hipMemcpyHostToDevice);
  hipMalloc(&cyc, 8 * 8);
  size_t lds = 5002 * 8;
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const char* names[] = {"fma_loop  (512 dependent mul+add)", "fma_flat  (4096 dependent mul+add, unrolled)", "lds       (256 round trips)", "table     (128 dependent shared reads)"};
  const double ops[] = {512, 4096, 256, 128};
  int sizes[] = {64, 256, 1024, 4096};
  for (int n : sizes) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemset(cyc, 0, 64);
      hipLaunchKernelGGL(probe, dim3(n), dim3(64), lds, 0, out, idx, cyc);
      hipDeviceSynchronize();
    }
    unsigned long long hc[8]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
    printf("blocks %d\n", n);
    for (int k = 0; k < 4; k++) printf("  %-48s %8.1f cycles/op\n", names[k], (double)hc[k] / n / ops[k]);
  }
  return 0;
}
