// Cost of one row step of the register PGS (mjrl_step.h) and of its parts, dependent chains, one wave per SIMD.
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
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc) {
  int L = threadIdx.x, kme = L & 15;
  double fi = out[L], r = out[L] + 0.25, ainv = 0.7, haii = 0.71, a3 = 0.01, a5 = 0.02, imp = 0;
  bool refused = false;
  long long t0, t1;
  // 1: row_newbcast alone, dependent (bcast + add)
  t0 = clock64();
  for (int i = 0; i < N; i++) { r = r * 0.999 + dpp<0x153>(r); }
  t1 = clock64(); if (L == 0) cyc[0] = t1 - t0;
  // 2: the unguarded step, two steps per iteration (lanes 3 and 5)
  t0 = clock64();
  for (int i = 0; i < N / 2; i++) {
    { double fn = fmax(fi - r * ainv, 0.0); double delta = fn - fi; double change = delta * delta * haii + delta * r;
      r += a3 * dpp<0x153>(delta); if (kme == 3) { fi = fn; imp -= change; refused |= change > 1e-10; } }
    { double fn = fmax(fi - r * ainv, 0.0); double delta = fn - fi; double change = delta * delta * haii + delta * r;
      r += a5 * dpp<0x155>(delta); if (kme == 5) { fi = fn; imp -= change; refused |= change > 1e-10; } }
  }
  t1 = clock64(); if (L == 0) cyc[1] = t1 - t0;
  // 3: fmax chain
  t0 = clock64();
  for (int i = 0; i < N; i++) r = fmax(r * 0.999, 0.001);
  t1 = clock64(); if (L == 0) cyc[2] = t1 - t0;
  // 4: compare + select chain
  t0 = clock64();
  for (int i = 0; i < N; i++) { double x = r * 0.999; if (x < 0.001) x = 0.001; r = x; }
  t1 = clock64(); if (L == 0) cyc[3] = t1 - t0;
  out[L] = r + fi + imp + refused;
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
  const char* names[] = {"row_newbcast f64 + mul + add (dependent)", "unguarded row step", "mul + fmax", "mul + compare + select"};
  for (int k = 0; k < 4; k++) printf("%-44s %8.1f cycles\n", names[k], (double)hc[k] / N);
  return 0;
}
