// Wave-level primitives for gfx950 (CDNA4): one 64-lane wavefront owns one env copy.
// Workgroups are exactly one wave (blockDim.x == 64), so the workgroup barrier is a wave barrier
// plus an LDS fence.  Everything cross-lane in the stepper goes through this header.
#ifndef MJRL_WAVE_H
#define MJRL_WAVE_H

#include <hip/hip_runtime.h>

namespace wv {

__device__ __forceinline__ int lane() { return threadIdx.x; }
__device__ __forceinline__ int env_index() { return blockIdx.x; }

// LDS writes of every lane become visible to every lane of the wave
__device__ __forceinline__ void sync() { __syncthreads(); }

__device__ __forceinline__ double shfl(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int shfl(int v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ unsigned long long ballot(bool pred) { return __ballot(pred); }
__device__ __forceinline__ int popc(unsigned long long x) { return __popcll(x); }

// all-reduce over the 64 lanes, xor butterfly (every lane ends with the same bits)
__device__ __forceinline__ double sum(double v) {
#pragma unroll
  for (int mask = 1; mask < 64; mask <<= 1) v += shfl_xor(v, mask);
  return v;
}
__device__ __forceinline__ double min_pos(double v) {   // minimum, every lane gets it
#pragma unroll
  for (int mask = 1; mask < 64; mask <<= 1) v = fmin(v, shfl_xor(v, mask));
  return v;
}

}  // namespace wv

#endif
