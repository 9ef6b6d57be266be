// Per-lane vector / quaternion / spatial-inertia arithmetic used by the stepper kernels.
// Spatial vectors are [angular(3); linear(3)]; composite inertias use the 10-number form
// [Ixx Iyy Izz Ixy Ixz Iyz  m*cx m*cy m*cz  m] about the kinematic tree's reference point in
// world orientation (the "c-frame" convention of MuJoCo's documentation).
#ifndef MJRL_MATH_H
#define MJRL_MATH_H

#include <math.h>

typedef double real;

#define MJ_MINVAL 1e-15
#define MJ_MINIMP 0.0001
#define MJ_MAXIMP 0.9999

namespace mj {

struct V3 {
  real x, y, z;
};

__host__ __device__ __forceinline__ V3 v3(real x, real y, real z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__host__ __device__ __forceinline__ V3 ld3(const real* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(real* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, real s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ real dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ real norm(V3 a) { return sqrt(dot(a, a)); }
// unit vector; a (near-)zero input becomes +x.  *len receives the original length.
__device__ __forceinline__ V3 normalized(V3 a, real* len) {
  real n = norm(a);
  if (len) *len = n;
  if (n < MJ_MINVAL) return v3(1, 0, 0);
  return a * (1.0 / n);
}

struct Quat {
  real w, x, y, z;
};
__host__ __device__ __forceinline__ Quat ldq(const real* p) { Quat q; q.w = p[0]; q.x = p[1]; q.y = p[2]; q.z = p[3]; return q; }
__device__ __forceinline__ void stq(real* p, Quat q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
__device__ __forceinline__ Quat qmul(Quat a, Quat b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return r;
}
__device__ __forceinline__ Quat qnormalized(Quat q) {
  real n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < MJ_MINVAL) { q.w = 1; q.x = q.y = q.z = 0; return q; }
  real s = 1.0 / n;
  q.w *= s; q.x *= s; q.y *= s; q.z *= s;
  return q;
}
__device__ __forceinline__ Quat axis_angle(V3 axis, real angle) {
  real s, c;
  sincos(0.5 * angle, &s, &c);      // one argument reduction for both
  Quat q; q.w = c; q.x = axis.x * s; q.y = axis.y * s; q.z = axis.z * s;
  return q;
}

// row-major rotation matrix
struct M3 {
  real m[9];
};
__device__ __forceinline__ M3 qmat(Quat q) {
  M3 r;
  real w = q.w, x = q.x, y = q.y, z = q.z;
  r.m[0] = w * w + x * x - y * y - z * z; r.m[1] = 2 * (x * y - w * z); r.m[2] = 2 * (x * z + w * y);
  r.m[3] = 2 * (x * y + w * z); r.m[4] = w * w - x * x + y * y - z * z; r.m[5] = 2 * (y * z - w * x);
  r.m[6] = 2 * (x * z - w * y); r.m[7] = 2 * (y * z + w * x); r.m[8] = w * w - x * x - y * y + z * z;
  return r;
}
__device__ __forceinline__ M3 ldm(const real* p) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.m[i] = p[i];
  return r;
}
__device__ __forceinline__ V3 mul(const M3& a, V3 v) {
  return v3(a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z);
}
__device__ __forceinline__ V3 mulT(const M3& a, V3 v) {
  return v3(a.m[0] * v.x + a.m[3] * v.y + a.m[6] * v.z, a.m[1] * v.x + a.m[4] * v.y + a.m[7] * v.z,
            a.m[2] * v.x + a.m[5] * v.y + a.m[8] * v.z);
}
__device__ __forceinline__ V3 col(const M3& a, int c) { return v3(a.m[c], a.m[3 + c], a.m[6 + c]); }
__device__ __forceinline__ V3 rot(Quat q, V3 v) { return mul(qmat(q), v); }

// ---- spatial algebra on plain arrays of 6 / 10
__device__ __forceinline__ void cross_motion(real* r, const real* vel, const real* v) {
  V3 a = cross(ld3(vel), ld3(v));
  V3 b = cross(ld3(vel), ld3(v + 3)) + cross(ld3(vel + 3), ld3(v));
  st3(r, a);
  st3(r + 3, b);
}
__device__ __forceinline__ void cross_force(real* r, const real* vel, const real* f) {
  V3 a = cross(ld3(vel), ld3(f)) + cross(ld3(vel + 3), ld3(f + 3));
  V3 b = cross(ld3(vel), ld3(f + 3));
  st3(r, a);
  st3(r + 3, b);
}
__device__ __forceinline__ void inert_mul(real* r, const real* i, const real* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
__device__ __forceinline__ real dot6(const real* a, const real* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}
// principal inertia `inert` with orientation `mat`, mass `mass`, centre offset `dif` from the reference point
__device__ __forceinline__ void inert_com(real* res, const real* inert, const M3& mat, V3 dif, real mass) {
  real t[9], full[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) t[3 * r + c] = mat.m[3 * r + c] * inert[c];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++)
      full[3 * r + c] = t[3 * r] * mat.m[3 * c] + t[3 * r + 1] * mat.m[3 * c + 1] + t[3 * r + 2] * mat.m[3 * c + 2];
  res[0] = full[0] + mass * (dif.y * dif.y + dif.z * dif.z);
  res[1] = full[4] + mass * (dif.x * dif.x + dif.z * dif.z);
  res[2] = full[8] + mass * (dif.x * dif.x + dif.y * dif.y);
  res[3] = full[1] - mass * dif.x * dif.y;
  res[4] = full[2] - mass * dif.x * dif.z;
  res[5] = full[5] - mass * dif.y * dif.z;
  res[6] = mass * dif.x; res[7] = mass * dif.y; res[8] = mass * dif.z;
  res[9] = mass;
}

// constraint impedance d(r) of the soft-constraint model (solimp = dmin dmax width midpoint power)
__device__ __forceinline__ real impedance(const real* solimp, real pos, real margin) {
  real dmin = fmin(fmax(solimp[0], MJ_MINIMP), MJ_MAXIMP);
  real dmax = fmin(fmax(solimp[1], MJ_MINIMP), MJ_MAXIMP);
  real width = fmax(solimp[2], MJ_MINVAL);
  real mid = fmin(fmax(solimp[3], MJ_MINIMP), MJ_MAXIMP);
  real power = fmax(solimp[4], 1.0);
  if (dmin == dmax || width <= MJ_MINVAL) return 0.5 * (dmin + dmax);
  real x = fabs((pos - margin) / width);
  if (x >= 1) return dmax;
  if (x == 0) return dmin;
  real y;
  if (power == 1) y = x;
  else if (power == 2) {            // the default exponent: a square, not a call to pow
    if (x <= mid) { real t = x / mid; y = t * t * mid; }
    else { real t = (1 - x) / (1 - mid); y = 1 - t * t * (1 - mid); }
  }
  else if (x <= mid) y = pow(x / mid, power) * mid;
  else y = 1 - pow((1 - x) / (1 - mid), power) * (1 - mid);
  return dmin + y * (dmax - dmin);
}

}  // namespace mj

#endif
