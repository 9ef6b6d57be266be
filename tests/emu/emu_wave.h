// TEST INFRASTRUCTURE: CPU stand-in for csrc/mjrl_wave.h.
// The 64 lanes of a wavefront run as 64 cooperative fibers on one thread; wv::sync() hands control to the
// next lane, so a lane resumes only after every other lane has reached the same barrier.  Cross-lane
// primitives go through a 64-slot exchange buffer with the same xor-butterfly order as the GPU header, so
// floating-point sums come out bit-identical to the device's.  The harness also counts barriers per lane:
// a kernel whose lanes disagree on the number of barriers (a divergent barrier on the GPU) fails the run.
#ifndef MJRL_WAVE_H
#define MJRL_WAVE_H

#include <cmath>
#include <cstddef>
#include <cstdint>

#define __device__
#define __host__
#define __forceinline__ inline

namespace emu {
extern int cur_lane, cur_env;
extern long sync_count[64];
extern double xd[64];
extern long long xi[64];
void yield_lane();
}  // namespace emu

namespace wv {
inline int lane() { return emu::cur_lane; }
inline int env_index() { return emu::cur_env; }
inline void sync() { emu::sync_count[emu::cur_lane]++; emu::yield_lane(); }
inline double shfl(double v, int src) {
  emu::xd[lane()] = v; sync();
  double r = emu::xd[src & 63]; sync();
  return r;
}
inline int shfl(int v, int src) {
  emu::xi[lane()] = v; sync();
  int r = (int)emu::xi[src & 63]; sync();
  return r;
}
inline double shfl_xor(double v, int mask) { return shfl(v, lane() ^ mask); }
inline unsigned long long ballot(bool pred) {
  emu::xi[lane()] = pred ? 1 : 0; sync();
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) if (emu::xi[i]) m |= (1ull << i);
  sync();
  return m;
}
inline int popc(unsigned long long x) { return __builtin_popcountll(x); }
inline unsigned long long clock() { return 0; }
inline unsigned long long realtime() { return 0; }
inline void atomic_add(unsigned long long* p, unsigned long long v) { *p += v; }
inline int atomic_add_int(int* p, int v) { int old = *p; *p += v; return old; }
inline void atomic_add_noret(int* p, int v) { *p += v; }
inline void atomic_or_noret(unsigned* p, unsigned v) { *p |= v; }
inline int popc32(unsigned x) { return __builtin_popcount(x); }
inline int first_set(unsigned long long x) { return __builtin_ctzll(x); }
inline double sum_n(double v, int width) {   // all-reduce over aligned groups of `width` lanes (16, 32 or 64)
  for (int mask = 1; mask < width; mask <<= 1) v += shfl_xor(v, mask);
  return v;
}
inline double sum(double v) { return sum_n(v, 64); }
inline int opaque_lane(int v) { return v; }
inline int opaque_uniform(int v) { return v; }
template <typename T>
inline const T* fresh(const T* p) { return p; }
inline void set_priority(int) {}
inline int first_int(int v) { return shfl(v, 0); }
inline double first(double v) { return shfl(v, 0); }   // lane 0's value in every lane
template <int N>
inline double bcast16(double v) { return shfl(v, (lane() & ~15) + N); }   // lane N of the caller's row of 16
template <int N>
inline int bcast16i(int v) { return shfl(v, (lane() & ~15) + N); }   // lane N of the caller's row of 16
inline double bcast16_var(double v, int n) { return shfl(v, (lane() & ~15) + (n & 15)); }
template <bool HIGH>
inline int half_to_all_i(int v) { return shfl(v, (lane() & 31) + (HIGH ? 32 : 0)); }
template <bool HIGH>
inline double half_to_all(double v) { return shfl(v, (lane() & 31) + (HIGH ? 32 : 0)); }
inline double sum16(double v) { return sum_n(v, 16); }
inline int lane_int(int v, int src) { return shfl(v, src); }
inline double lane_value(double v, int src) { return shfl(v, src); }
inline double rows4_sum(double v) { return (shfl(v, 0) + shfl(v, 16)) + (shfl(v, 32) + shfl(v, 48)); }
inline double rows_sum(double v, int rows) {
  if (rows <= 1) return shfl(v, 0);
  if (rows == 2) return shfl(v, 0) + shfl(v, 16);
  return rows4_sum(v);
}
inline double min_pos(double v) {
  for (int mask = 1; mask < 64; mask <<= 1) v = std::fmin(v, shfl_xor(v, mask));
  return v;
}
}  // namespace wv

#endif
