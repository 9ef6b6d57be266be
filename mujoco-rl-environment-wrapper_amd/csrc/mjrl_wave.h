// Wave-level primitives for gfx950 (CDNA4): one 64-lane wavefront owns one env copy.
// Workgroups are exactly one wave (blockDim.x == 64), so the workgroup barrier is a wave barrier
// plus an LDS fence.  Everything cross-lane in the stepper goes through this header.
#ifndef MJRL_WAVE_H
#define MJRL_WAVE_H

#include <hip/hip_runtime.h>

namespace wv {

__device__ __forceinline__ int lane() { return threadIdx.x; }
__device__ __forceinline__ int env_index() { return blockIdx.x; }

// LDS writes of every lane become visible to every lane of the wave.  The workgroup IS one wave, and the LDS
// executes one wave's DS instructions in issue order, so no s_barrier and no counter drain are needed: a
// wavefront-scope fence keeps the compiler from moving LDS accesses across this point and emits no instruction,
// which leaves global loads issued earlier (table prefetches) in flight instead of draining them at every stage
// boundary the way __syncthreads() (s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier) would.
__device__ __forceinline__ void sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double shfl(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int shfl(int v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ double shfl_xor(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ unsigned long long ballot(bool pred) { return __ballot(pred); }
__device__ __forceinline__ int popc(unsigned long long x) { return __popcll(x); }
// wave clock (s_memtime) and a device-scope counter add, for the diagnostic stage stamps only
__device__ __forceinline__ unsigned long long clock() { return (unsigned long long)clock64(); }
// constant-rate counter shared by every CU (100 MHz): comparable between waves, unlike the shader clock
__device__ __forceinline__ unsigned long long realtime() { return (unsigned long long)wall_clock64(); }
__device__ __forceinline__ void atomic_add(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
// device-scope counter add returning the old value (work-bucket slots of the longest-first dispatch)
__device__ __forceinline__ int atomic_add_int(int* p, int v) { return atomicAdd(p, v); }

// device-scope counter / bit-set updates whose old value nobody needs: no register comes back, so the wave never waits
// for them (an atomic WITH a return value is waited for where it is issued, and 2048 waves queueing on a handful of
// addresses made that wait the longest stall of the whole step)
__device__ __forceinline__ void atomic_add_noret(int* p, int v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_or_noret(unsigned* p, unsigned v) {
  (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int popc32(unsigned x) { return __builtin_popcount(x); }
__device__ __forceinline__ int first_set(unsigned long long x) { return __builtin_ctzll(x); }    // x != 0

// 64-bit value moved with a DPP control word (two v_mov_b32_dpp); every lane reads a lane of its own row of 16
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  // every control word used here reads a lane that exists (permutations inside a row of 16, row_newbcast), so the
  // "old" operand never shows: passing 0 with bound_ctrl spares the two copies of the source a tied operand costs
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// All-reduce (sum) over aligned groups of `width` lanes, width in {16, 32, 64}: every lane ends with its group's
// sum.  Inside a row of 16 the partner exchange is pure DPP (quad_perm xor 1, quad_perm xor 2, row_half_mirror,
// row_mirror); each step pairs lanes that already hold equal partial sums, so the result has the bits of an
// xor butterfly.  Rows are then combined through the LDS crossbar (ds_bpermute).
__device__ __forceinline__ double sum_n(double v, int width) {
  v += dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp<0x141>(v);   // row_half_mirror
  v += dpp<0x140>(v);   // row_mirror
  if (width > 16) v += shfl_xor(v, 16);
  if (width > 32) v += shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ double sum(double v) { return sum_n(v, 64); }

// value of lane N of the caller's row of 16 lanes, in every lane of that row (DPP row_newbcast, no LDS traffic)
// (ONE v_mov_b64_dpp: row_newbcast is the control word the 64-bit DPP move of gfx90a and later takes, and the builtin
// forms it from a double operand; the other controls stay two 32-bit moves.  35 against 38 cycles per row step of the
// register solvers in isolation, tools/dpp64_probe.hip -- 6 % of the long solves the launch waits for.)
// (In the model-specialised kernels only: with it the generic kernels of libmjrl_hip.so -- every size a run-time value --
// run into a code generator error of this compiler, "Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_shared_base".
// The same bits either way.)
template <int N>
__device__ __forceinline__ double bcast16(double v) {
#if defined(MJRL_SPEC)
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + N, 0xF, 0xF, true);
#else
  return dpp<0x150 + N>(v);
#endif
}

template <int N>
__device__ __forceinline__ int bcast16i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xF, 0xF, true); }
// the same with the lane as an argument: for fully unrolled loops, where it is a constant by the time the switch is seen
__device__ __forceinline__ double bcast16_var(double v, int n) {
  switch (n & 15) {
    case 0: return bcast16<0>(v); case 1: return bcast16<1>(v); case 2: return bcast16<2>(v); case 3: return bcast16<3>(v);
    case 4: return bcast16<4>(v); case 5: return bcast16<5>(v); case 6: return bcast16<6>(v); case 7: return bcast16<7>(v);
    case 8: return bcast16<8>(v); case 9: return bcast16<9>(v); case 10: return bcast16<10>(v); case 11: return bcast16<11>(v);
    case 12: return bcast16<12>(v); case 13: return bcast16<13>(v); case 14: return bcast16<14>(v); default: return bcast16<15>(v);
  }
}

// the caller's value from the lower (HIGH = false) or upper half of the wave, in both halves: lane i and lane i + 32
// both get lane (i + 32 HIGH)'s value.  One v_permlane32_swap per dword (gfx950).
template <bool HIGH>
__device__ __forceinline__ int half_to_all_i(int v) {
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return HIGH ? r[1] : r[0];
}
template <bool HIGH>
__device__ __forceinline__ double half_to_all(double v) {
  return __hiloint2double(half_to_all_i<HIGH>(__double2hiint(v)), half_to_all_i<HIGH>(__double2loint(v)));
}

// all-reduce (sum) inside each row of 16 lanes only
__device__ __forceinline__ double sum16(double v) { return sum_n(v, 16); }

// sum of the values held by lanes 0, 16, 32 and 48 (one per row of 16), in every lane: v_readlane, no LDS traffic.
// Association (r0 + r1) + (r2 + r3), the same as the xor butterfly gives for a value that is zero elsewhere.
__device__ __forceinline__ double lane_value(double v, int src) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// lane `src`'s int in every lane; src is wave-uniform (v_readlane)
__device__ __forceinline__ int lane_int(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double rows4_sum(double v) {
  return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
// the same when only the first `rows` rows of 16 lanes can hold non-zero values (x + 0 is exact, so the sums agree
// bit for bit); `rows` is wave-uniform, a constant in the model-specialised kernel
__device__ __forceinline__ double rows_sum(double v, int rows) {
  if (rows <= 1) return lane_value(v, 0);
  if (rows == 2) return lane_value(v, 0) + lane_value(v, 16);
  return rows4_sum(v);
}

// lane 0's value in every lane (v_readfirstlane: the result is wave-uniform)
__device__ __forceinline__ double first(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// The same value with its origin hidden from the optimiser (no instruction).  Used at the top of a hot loop on the
// values its per-step conditions derive from: hoisted out of the loop those conditions become one 64-bit lane mask
// each, too many for the scalar registers, and come back through v_readlane at every use.
__device__ __forceinline__ int opaque_lane(int v) { asm volatile("" : "+v"(v)); return v; }
// (through a VECTOR register and back with v_readfirstlane: an "s" constraint on the asm operand makes the build fail --
// "illegal VGPR to SGPR copy" -- whenever the register allocator, short of scalar registers in a kernel this size, has
// chosen to keep the value in a vector register at that point; a copy into a vector register is always legal)
__device__ __forceinline__ int opaque_uniform(int v) {
  asm volatile("" : "+v"(v));
  return __builtin_amdgcn_readfirstlane(v);
}
// The same (wave-uniform) pointer INTO CONSTANT MEMORY (the kernel's argument segment) as a value the optimiser has not
// seen before: what is loaded through it is not merged with earlier loads through the original, so nothing fetched early
// stays live in registers until here.  The result is re-typed as a constant-address-space pointer before it goes back
// to a generic one: the loads through it stay scalar loads (through a pointer of unknown origin they would be flat
// vector loads, every lane fetching the same word).
template <typename T>
__device__ __forceinline__ const T* fresh(const T* p) {
  unsigned long long v = (unsigned long long)p;
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  asm volatile("" : "+v"(lo), "+v"(hi));       // (see opaque_uniform)
  lo = (unsigned)__builtin_amdgcn_readfirstlane((int)lo);
  hi = (unsigned)__builtin_amdgcn_readfirstlane((int)hi);
  return (const T*)(const T __attribute__((address_space(4)))*)(((unsigned long long)hi << 32) | lo);
}

// issue priority of this wave among the waves of its SIMD (0 lowest .. 3 highest); p is wave-uniform
__device__ __forceinline__ void set_priority(int p) {
  if (p >= 3) __builtin_amdgcn_s_setprio(3);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// lane 0's int in a scalar register: tells the compiler that a value it could not prove wave-uniform is uniform, so
// that conditions on it become scalar branches instead of per-lane masks
__device__ __forceinline__ int first_int(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ double min_pos(double v) {   // minimum over the wave, every lane gets it
  // inside the rows of 16 by DPP, across the four rows by v_readlane (the minimum is exact: any order gives the same)
  v = fmin(v, dpp<0xB1>(v));
  v = fmin(v, dpp<0x4E>(v));
  v = fmin(v, dpp<0x141>(v));
  v = fmin(v, dpp<0x140>(v));
  return fmin(fmin(lane_value(v, 0), lane_value(v, 16)), fmin(lane_value(v, 32), lane_value(v, 48)));
}

}  // namespace wv

#endif
