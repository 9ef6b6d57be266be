// The register solver's row step (mjrl_step.h: max, broadcast, multiply, subtract) with the broadcast as two
// v_mov_b32_dpp (what the compiler makes of the 32-bit builtin) against one v_mov_b64_dpp row_newbcast.
// One wave per SIMD, dependent chain, cycles per step; also checks that both give the same bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int K>
__device__ __forceinline__ double bc32(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ double bc64(double v) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
  return r;
}
#define STEP(BC, KK, AK) { double db = BC<KK>(fmax(ns, nf)); if (kme == KK) nss = ns; ns -= AK * db; }
__global__ __launch_bounds__(64) void probe(double* out, long long* cyc) {
  int L = threadIdx.x;
  int kme = L & 15;
  asm volatile("" : "+v"(kme));
  double nf = -out[L], a3 = 0.01, a5 = 0.02, a7 = 0.015, a9 = 0.005;
  long long t0, t1;
  double ns = out[L] + 0.25, nss = 0;
  t0 = clock64();
  for (int i = 0; i < N / 4; i++) { STEP(bc32, 3, a3) STEP(bc32, 5, a5) STEP(bc32, 7, a7) STEP(bc32, 9, a9) }
  t1 = clock64(); if (L == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  double r32 = ns + nss;
  ns = out[L] + 0.25; nss = 0;
  t0 = clock64();
  for (int i = 0; i < N / 4; i++) { STEP(bc64, 3, a3) STEP(bc64, 5, a5) STEP(bc64, 7, a7) STEP(bc64, 9, a9) }
  t1 = clock64(); if (L == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
  double r64 = ns + nss;
  if (blockIdx.x == 0) { out[64 + L] = r32; out[128 + L] = r64; }
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 192 * 8);
  double h[192]; for (int i = 0; i < 192; i++) h[i] = 0.5 + 0.01 * (i % 64);
  hipMemcpy(out, h, sizeof(h), hipMemcpyHostToDevice);
  hipMalloc(&cyc, 8 * 8);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(probe, dim3(1024), dim3(64), 0, 0, out, cyc);
  hipDeviceSynchronize();
  long long hc[8]; hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost);
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int same = 1; for (int i = 0; i < 64; i++) same &= (h[64 + i] == h[128 + i]);
  printf("row step, two v_mov_b32_dpp: %6.1f cycles\nrow step, one v_mov_b64_dpp: %6.1f cycles\nsame bits: %s\n",
         (double)hc[0] / N, (double)hc[1] / N, same ? "yes" : "NO");
  return 0;
}
