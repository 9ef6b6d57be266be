// Calibration of the step kernel's cost model on gfx950: dependent-chain latencies with ONE wave per SIMD
// (the kernel's regime).  Prints cycles per operation (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
__device__ __forceinline__ double dpp_sum16(double v) {
  auto mv = [](double x, auto ctrl) { return x; };
  (void)mv;
  int lo, hi;
#define STEP(CTRL)                                                                 \
  lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xF, 0xF, false); \
  hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xF, 0xF, false); \
  v += __hiloint2double(hi, lo);
  STEP(0xB1) STEP(0x4E) STEP(0x141) STEP(0x140)
#undef STEP
  return v;
}
__global__ __launch_bounds__(64) void probe(double* out, const int* idx, long long* cyc) {
  __shared__ double lds[4096];
  int L = threadIdx.x;
  double x = out[L] + 1.0, y = 1.000001;
  long long t0, t1;
  // 1: dependent fma chain
  t0 = clock64();
  for (int i = 0; i < N; i++) x = x * y + 0.5;
  t1 = clock64(); if (L == 0) cyc[0] = (t1 - t0);
  // 2: dependent division chain
  t0 = clock64();
  for (int i = 0; i < N; i++) x = 1.0 / (x + 2.0);
  t1 = clock64(); if (L == 0) cyc[1] = (t1 - t0);
  // 3: LDS write -> read by another lane (dependent)
  lds[L] = x;
  t0 = clock64();
  for (int i = 0; i < N; i++) { lds[L] = x; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); x = lds[(L + 1) & 63] + 1.0; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
  t1 = clock64(); if (L == 0) cyc[2] = (t1 - t0);
  // 4: DPP 16-lane all-reduce, dependent
  t0 = clock64();
  for (int i = 0; i < N; i++) x = dpp_sum16(x) * 1e-3;
  t1 = clock64(); if (L == 0) cyc[3] = (t1 - t0);
  // 5: dependent global load chain (pointer chasing in an L2-resident table)
  int j = L;
  t0 = clock64();
  for (int i = 0; i < N; i++) j = idx[j];
  t1 = clock64(); if (L == 0) cyc[4] = (t1 - t0);
  // 6: sqrt chain
  t0 = clock64();
  for (int i = 0; i < N; i++) x = sqrt(x + 3.0);
  t1 = clock64(); if (L == 0) cyc[5] = (t1 - t0);
  // 7: sincos chain
  double s, c;
  t0 = clock64();
  for (int i = 0; i < N; i++) { sincos(x, &s, &c); x = s + c; }
  t1 = clock64(); if (L == 0) cyc[6] = (t1 - t0);
  // 8: ds_bpermute shfl_xor of a double, dependent
  t0 = clock64();
  for (int i = 0; i < N; i++) x += __shfl_xor(x, 16, 64);
  t1 = clock64(); if (L == 0) cyc[7] = (t1 - t0);
  // 9: independent LDS reads (8 in flight) then sum
  t0 = clock64();
  for (int i = 0; i < N; i++) { double a = 0; for (int k = 0; k < 8; k++) a += lds[(L + k * 7 + i) & 4095]; x += a; }
  t1 = clock64(); if (L == 0) cyc[8] = (t1 - t0);
  out[L] = x + j;
}
int main() {
  double* out; int* idx; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMemset(out, 0, 64 * 8);
  int h[4096]; for (int i = 0; i < 4096; i++) h[i] = (i * 37 + 11) & 4095;
  hipMalloc(&idx, sizeof(h)); hipMemcpy(idx, h, sizeof(h), hipMemcpyHostToDevice);
  hipMalloc(&cyc, 16 * 8);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, out, idx, cyc);
  hipDeviceSynchronize();
  long long hc[16]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  const char* names[] = {"fma f64 (dependent)", "div f64 (dependent, incl. add)", "LDS write->other-lane read round trip", "DPP sum16 f64 (+mul)",
                         "global load, dependent (L1/L2 hit)", "sqrt f64 (+add)", "sincos f64 (+add)", "shfl_xor f64 via ds_bpermute (+add)", "8 independent LDS reads + adds"};
  for (int k = 0; k < 9; k++) printf("%-46s %8.1f cycles\n", names[k], (double)hc[k] / N);
  return 0;
}
