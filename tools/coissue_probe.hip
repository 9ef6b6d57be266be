// Do instructions of different kinds from the two waves of a SIMD overlap?  8 one-wave workgroups per CU (sized by LDS, like
// the step kernel), the wave in slot 0 of a SIMD runs kind A, the wave in slot 1 kind B; time of the pair against each
// kind alone (the other wave idle-spinning on s_sleep).  Kinds: 0 dependent v_fma_f64 chain, 1 s_add chain, 2 ds_read
// chain, 3 independent v_fma_f64 (4 chains), 4 idle.  hipcc -O2 --offload-arch=gfx950 -o /tmp/coissue tools/coissue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 20000

__device__ __forceinline__ unsigned hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}

template <int KIND>
__device__ __forceinline__ double run_kind(double x, double* lds) {
  if constexpr (KIND == 0) {
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) x = __builtin_fma(x, 1.0000001, 1e-9);
    }
  } else if constexpr (KIND == 1) {
    unsigned s = 1;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
    }
    x += s;
  } else if constexpr (KIND == 2) {
    int idx = threadIdx.x;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) { double v = lds[idx]; idx = (idx + (int)(v != 12345.0)) & 63; }
    }
    x += idx;
  } else if constexpr (KIND == 3) {
    double y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3;
    for (int i = 0; i < N_ITER; i++) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        y0 = __builtin_fma(y0, 1.0000001, 1e-9); y1 = __builtin_fma(y1, 1.0000001, 1e-9);
        y2 = __builtin_fma(y2, 1.0000001, 1e-9); y3 = __builtin_fma(y3, 1.0000001, 1e-9);
      }
    }
    x = y0 + y1 + y2 + y3;
  }
  return x;
}

template <int A, int B>
__global__ __launch_bounds__(64) void probe(double* out, unsigned long long* cyc, unsigned* ids) {
  extern __shared__ double lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned id = hw_id();
  const unsigned slot = id & 15u;
  const unsigned long long t0 = clock64();
  double x = threadIdx.x;
  if ((slot & 1u) == 0) x = run_kind<A>(x, lds); else x = run_kind<B>(x, lds);
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = x;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; ids[blockIdx.x] = id; }
}

template <int A, int B>
void run(const char* name, double* out, unsigned long long* cyc, unsigned* ids, int n_wg) {
  hipLaunchKernelGGL((probe<A, B>), dim3(n_wg), dim3(64), 19 * 1024, 0, out, cyc, ids);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(n_wg);
  std::vector<unsigned> hi(n_wg);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * n_wg, hipMemcpyDeviceToHost);
  hipMemcpy(hi.data(), ids, sizeof(unsigned) * n_wg, hipMemcpyDeviceToHost);
  double s[2] = {0, 0}; int c[2] = {0, 0};
  for (int i = 0; i < n_wg; i++) { int r = hi[i] & 1; s[r] += (double)h[i]; c[r]++; }
  printf("%-28s slot-even waves (kind %d): %8.0f cycles (%d waves)   slot-odd waves (kind %d): %8.0f cycles (%d waves)   per 8 instr: %.1f / %.1f\n",
         name, A, c[0] ? s[0] / c[0] : 0, c[0], B, c[1] ? s[1] / c[1] : 0, c[1], c[0] ? s[0] / c[0] / N_ITER : 0, c[1] ? s[1] / c[1] / N_ITER : 0);
}

int main() {
  const int n_wg = 256 * 8;
  double* out; unsigned long long* cyc; unsigned* ids;
  hipMalloc(&out, sizeof(double) * 64 * n_wg); hipMalloc(&cyc, sizeof(unsigned long long) * n_wg); hipMalloc(&ids, sizeof(unsigned) * n_wg);
  run<0, 4>("fma chain | idle", out, cyc, ids, n_wg);
  run<0, 0>("fma chain | fma chain", out, cyc, ids, n_wg);
  run<3, 4>("4 fma chains | idle", out, cyc, ids, n_wg);
  run<3, 3>("4 fma chains | 4 fma chains", out, cyc, ids, n_wg);
  run<1, 4>("salu | idle", out, cyc, ids, n_wg);
  run<0, 1>("fma chain | salu", out, cyc, ids, n_wg);
  run<3, 1>("4 fma chains | salu", out, cyc, ids, n_wg);
  run<2, 4>("lds chain | idle", out, cyc, ids, n_wg);
  run<0, 2>("fma chain | lds chain", out, cyc, ids, n_wg);
  run<3, 2>("4 fma chains | lds chain", out, cyc, ids, n_wg);
  run<1, 1>("salu | salu", out, cyc, ids, n_wg);
  return 0;
}
