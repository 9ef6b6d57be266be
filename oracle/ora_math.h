/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C fp64 restatement of the rigid-body step the reference obtains from the third-party
 * package mujoco==2.3.3 (requirements.txt:45; call sites mujoco_parent.py:126,335,349-350).
 * That package's source is NOT under /root/reference and is not installed in this image, so the
 * arithmetic here is written from MuJoCo's published algorithm description (documentation
 * "Computation" chapter: kinematics, CRB inertia, RNE bias, soft-constraint model with
 * solref/solimp impedance, pyramidal friction cones, PGS solver, semi-implicit Euler).
 *
 * PARITY UNPINNED for physics: no reference test or fixture asserts a qpos/qvel/sensordata
 * value, and the real library cannot be run here.  The oracle is pinned
 * by analytic known-answer tests only (tests/test_oracle_physics.py).
 *
 * This header: small vector / quaternion / spatial-algebra helpers.
 */
#ifndef ORA_MATH_H
#define ORA_MATH_H

#include <math.h>
#include <string.h>

#define ORA_MINVAL 1e-15
#define ORA_PI 3.14159265358979323846

static inline void v3_copy(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void v3_zero(double* r) { r[0] = r[1] = r[2] = 0.0; }
static inline void v3_add(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static inline void v3_sub(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void v3_scl(double* r, const double* a, double s) { r[0] = a[0] * s; r[1] = a[1] * s; r[2] = a[2] * s; }
static inline void v3_addscl(double* r, const double* a, const double* b, double s) { r[0] = a[0] + b[0] * s; r[1] = a[1] + b[1] * s; r[2] = a[2] + b[2] * s; }
static inline double v3_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline double v3_norm(const double* a) { return sqrt(v3_dot(a, a)); }
static inline void v3_cross(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
/* normalise in place; a (near-)zero vector becomes +x; returns the original length */
static inline double v3_normalize(double* a) {
  double n = v3_norm(a);
  if (n < ORA_MINVAL) { a[0] = 1.0; a[1] = 0.0; a[2] = 0.0; }
  else { double s = 1.0 / n; a[0] *= s; a[1] *= s; a[2] *= s; }
  return n;
}

/* r = M v, M row-major 3x3 */
static inline void m3_mulv(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  double y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  double z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
/* r = M^T v */
static inline void m3_mulTv(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  double y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  double z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}

static inline void q_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static inline void q_normalize(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < ORA_MINVAL) { q[0] = 1.0; q[1] = q[2] = q[3] = 0.0; }
  else { double s = 1.0 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
static inline void q_to_mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static inline void q_rotv(double* r, const double* q, const double* v) {
  double m[9];
  q_to_mat(m, q);
  m3_mulv(r, m, v);
}
static inline void q_axis_angle(double* q, const double* axis, double angle) {
  double s = sin(0.5 * angle);
  q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* ---- 6-D spatial vectors: [angular(3); linear(3)] ; inertia in the 10-number com-based form
 * [Ixx Iyy Izz Ixy Ixz Iyz  m*cx m*cy m*cz  m] about the tree's reference point, world orientation */
static inline void sp_cross_motion(double* r, const double* vel, const double* v) {
  double a[3], b[3], c[3];
  v3_cross(a, vel, v);
  v3_cross(b, vel, v + 3);
  v3_cross(c, vel + 3, v);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static inline void sp_cross_force(double* r, const double* vel, const double* f) {
  double a[3], b[3], c[3];
  v3_cross(a, vel, f);
  v3_cross(b, vel + 3, f + 3);
  v3_cross(c, vel, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
static inline void sp_inert_mulv(double* r, const double* i, const double* v) {
  r[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2] - i[8] * v[4] + i[7] * v[5];
  r[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2] + i[8] * v[3] - i[6] * v[5];
  r[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2] - i[7] * v[3] + i[6] * v[4];
  r[3] = i[8] * v[1] - i[7] * v[2] + i[9] * v[3];
  r[4] = i[6] * v[2] - i[8] * v[0] + i[9] * v[4];
  r[5] = i[7] * v[0] - i[6] * v[1] + i[9] * v[5];
}
static inline double sp_dot(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

#endif
