// Cost of the per-sweep tail of the register PGS: cost change of the lane's row, sum over the wave, convergence branch.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 1024
template <int CTRL>
__device__ __forceinline__ double dpp_copy(double v) {      // as in mjrl_wave.h: old == src
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_fresh(double v) {     // old = 0, bound_ctrl: no copy of the source needed
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc) {
  int L = threadIdx.x;
  double fi = out[L], f0 = fi * 0.99, rs = 0.3, haii = 0.7, scale = 1e-3, tol = -1e300;
  long long t0, t1;
  int iters = 0;
  // 1: the tail as it is: sum16 (copying dpp) + two readlanes + compare + loop branch
  t0 = clock64();
  for (int i = 0; i < N; i++) {
    double d = fi - f0, c = d * d * haii + d * rs, v = -c;
    bool refused = c > 1e-10;
    if (__ballot(refused)) fi *= 0.5;
    v += dpp_copy<0xB1>(v); v += dpp_copy<0x4E>(v); v += dpp_copy<0x141>(v); v += dpp_copy<0x140>(v);
    double imp = lane_value(v, 0) + lane_value(v, 16);
    iters++;
    fi = fi * 0.999 + 1e-9;
    if (imp * scale < tol) break;
  }
  t1 = clock64(); if (L == 0) cyc[0] = t1 - t0;
  // 2: fresh-destination dpp
  t0 = clock64();
  for (int i = 0; i < N; i++) {
    double d = fi - f0, c = d * d * haii + d * rs, v = -c;
    bool refused = c > 1e-10;
    if (__ballot(refused)) fi *= 0.5;
    v += dpp_fresh<0xB1>(v); v += dpp_fresh<0x4E>(v); v += dpp_fresh<0x141>(v); v += dpp_fresh<0x140>(v);
    double imp = lane_value(v, 0) + lane_value(v, 16);
    iters++;
    fi = fi * 0.999 + 1e-9;
    if (imp * scale < tol) break;
  }
  t1 = clock64(); if (L == 0) cyc[1] = t1 - t0;
  // 3: no reduction at all (lower bound: change + branches)
  t0 = clock64();
  for (int i = 0; i < N; i++) {
    double d = fi - f0, c = d * d * haii + d * rs, v = -c;
    bool refused = c > 1e-10;
    if (__ballot(refused)) fi *= 0.5;
    iters++;
    fi = fi * 0.999 + 1e-9;
    if (v * scale < tol - 1.0) break;
  }
  t1 = clock64(); if (L == 0) cyc[2] = t1 - t0;
  out[L] = fi + iters;
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
  const char* names[] = {"sweep tail as is", "sweep tail, dpp without source copies", "sweep tail without the reduction"};
  for (int k = 0; k < 3; k++) printf("%-44s %8.1f cycles\n", names[k], (double)hc[k] / N);
  return 0;
}
