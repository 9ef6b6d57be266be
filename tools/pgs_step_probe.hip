// Cost of one row step of the register PGS (mjrl_step.h) and of its parts: dependent chains, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 1024
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
#define STEP_FULL(KK, AK) { double fn = fmax(fi - r * ainv, 0.0); double db = dpp<0x150 + KK>(fn - fi); if (kme == KK) { fi = fn; rs = r; } r += AK * db; }
#define STEP_NORS(KK, AK) { double fn = fmax(fi - r * ainv, 0.0); double db = dpp<0x150 + KK>(fn - fi); if (kme == KK) { fi = fn; } r += AK * db; }
#define STEP_BARE(KK, AK) { double fn = fmax(fi - r * ainv, 0.0); double db = dpp<0x150 + KK>(fn - fi); fi = fn; r += AK * db; }
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc) {
  int L = threadIdx.x;
  int kme = L & 15;
  asm volatile("" : "+v"(kme));
  double fi = out[L], r = out[L] + 0.25, ainv = 0.7, a3 = 0.01, a5 = 0.02, a7 = 0.015, a9 = 0.005, rs = 0;
  long long t0, t1;
  t0 = clock64();
  for (int i = 0; i < N / 4; i++) { STEP_FULL(3, a3) STEP_FULL(5, a5) STEP_FULL(7, a7) STEP_FULL(9, a9) }
  t1 = clock64(); if (L == 0) cyc[0] = t1 - t0;
  t0 = clock64();
  for (int i = 0; i < N / 4; i++) { STEP_NORS(3, a3) STEP_NORS(5, a5) STEP_NORS(7, a7) STEP_NORS(9, a9) }
  t1 = clock64(); if (L == 0) cyc[1] = t1 - t0;
  t0 = clock64();
  for (int i = 0; i < N / 4; i++) { STEP_BARE(3, a3) STEP_BARE(5, a5) STEP_BARE(7, a7) STEP_BARE(9, a9) }
  t1 = clock64(); if (L == 0) cyc[2] = t1 - t0;
  // with the early-exit test of the real sweep in front of every step (tmax uniform, unknown to the compiler)
  int tmax = (int)out[0] + 4;
  tmax = __builtin_amdgcn_readfirstlane(tmax);
  t0 = clock64();
  for (int i = 0; i < N / 4; i++) {
    do {
      if (0 >= tmax) break; STEP_FULL(3, a3)
      if (1 >= tmax) break; STEP_FULL(5, a5)
      if (2 >= tmax) break; STEP_FULL(7, a7)
      if (3 >= tmax) break; STEP_FULL(9, a9)
      if (4 >= tmax) break; STEP_FULL(10, a9)
      if (5 >= tmax) break; STEP_FULL(11, a9)
    } while (0);
  }
  t1 = clock64(); if (L == 0) cyc[4] = t1 - t0;
  // the u-form step of the pipelined path for comparison: 16-lane reduction per row
  double u = fi, bid = 0.3, dinv = 0.9;
  t0 = clock64();
  for (int i = 0; i < N; i++) {
    double v = bid * dinv * u;
    v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
    double res = v + 0.1 * fi + 0.01;
    double fn = fmax(fi - res * ainv, 0.0);
    double delta = fn - fi; fi = fn; u += delta * bid;
  }
  t1 = clock64(); if (L == 0) cyc[3] = t1 - t0;
  out[L] = r + fi + rs + u;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8);
  double h[64]; for (int i = 0; i < 64; i++) h[i] = 0.5 + 0.01 * i;
  hipMemcpy(out, h, sizeof(h), hipMemcpyHostToDevice);
  hipMalloc(&cyc, 8 * 8);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, out, cyc);
  hipDeviceSynchronize();
  long long hc[8]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  const char* names[] = {"row step (captures f and r in the row's lane)", "row step, f captured only", "row step, no capture (chain only)", "u-form step (sum16 reduction)", "row step behind an early-exit test (4 of 6 run)"};
  for (int k = 0; k < 5; k++) printf("%-48s %8.1f cycles\n", names[k], (double)hc[k] / N);
  return 0;
}
