// Single-precision ray casts for the camera kernel (mjrl_render_kernel): the same intersection routines as the
// rangefinder's fp64 ones in mjrl_collide.h (ray_geom, geom_normal), in float.  An image is 8-bit: a ray that grazes a
// silhouette may fall on the other side of it than the fp64 oracle's ray caster says (tests allow 0.2 % of the pixels),
// nothing else changes -- and fp32 vector arithmetic runs at twice the fp64 rate with half the registers.
#ifndef MJRL_RAYF_H
#define MJRL_RAYF_H

#include "mjrl_collide.h"

namespace mj {

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 ldf3(const float* p) { return f3(p[0], p[1], p[2]); }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dotf(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 normalizedf(F3 a) {
  float n = sqrtf(dotf(a, a));
  return n < 1e-20f ? f3(1, 0, 0) : a * (1.0f / n);
}
// row-major 3x3 in nine floats
__device__ __forceinline__ F3 mulf(const float* m, F3 v) {
  return f3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
__device__ __forceinline__ F3 mulTf(const float* m, F3 v) {
  return f3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z);
}
__device__ __forceinline__ F3 colf(const float* m, int c) { return f3(m[c], m[3 + c], m[6 + c]); }

__device__ __forceinline__ float ray_sphere_atf(F3 center, float r, F3 pnt, F3 vec) {
  F3 rel = pnt - center;
  float b = dotf(vec, rel), cc = dotf(rel, rel) - r * r;
  float det = b * b - cc;
  if (det < 0) return -1;
  float sq = sqrtf(det);
  float x0 = -b - sq, x1 = -b + sq;
  if (x0 >= 0) return x0;
  if (x1 >= 0) return x1;
  return -1;
}

__device__ __forceinline__ float ray_geomf(int type, F3 gp, const float* gm, F3 gs, F3 pnt, F3 vec) {
  F3 rel = pnt - gp;
  if (type == GEOM_PLANE) {
    F3 n = colf(gm, 2);
    float denom = dotf(vec, n);
    if (denom > -1e-15f) return -1;
    float x = -dotf(rel, n) / denom;
    if (x < 0) return -1;
    F3 hit = rel + vec * x;
    if (gs.x > 0 && fabsf(dotf(hit, colf(gm, 0))) > gs.x) return -1;
    if (gs.y > 0 && fabsf(dotf(hit, colf(gm, 1))) > gs.y) return -1;
    return x;
  }
  if (type == GEOM_SPHERE) return ray_sphere_atf(gp, gs.x, pnt, vec);
  if (type == GEOM_CAPSULE) {
    F3 axis = colf(gm, 2);
    float r = gs.x, len = gs.y, best = -1;
    float va = dotf(vec, axis), ra = dotf(rel, axis);
    F3 vp = vec - axis * va, rp = rel - axis * ra;
    float a = dotf(vp, vp), b = dotf(vp, rp), cc = dotf(rp, rp) - r * r;
    if (a > 1e-15f) {
      float det = b * b - a * cc;
      if (det >= 0) {
        float sq = sqrtf(det), inv = 1.0f / a;
        float xs[2] = {(-b - sq) * inv, (-b + sq) * inv};
#pragma unroll
        for (int k = 0; k < 2; k++) {
          float x = xs[k];
          if (x >= 0 && fabsf(ra + x * va) <= len && (best < 0 || x < best)) best = x;
        }
      }
    }
#pragma unroll
    for (int s = -1; s <= 1; s += 2) {
      float x = ray_sphere_atf(gp + axis * (s * len), r, pnt, vec);
      if (x >= 0) {
        float h = ra + x * va;
        if (s * h >= len && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  if (type == GEOM_BOX) {
    F3 lpv = mulTf(gm, rel), lvv = mulTf(gm, vec);
    float lp[3] = {lpv.x, lpv.y, lpv.z}, lv[3] = {lvv.x, lvv.y, lvv.z}, s[3] = {gs.x, gs.y, gs.z};
    float best = -1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (fabsf(lv[k]) < 1e-15f) continue;
      const float inv = 1.0f / lv[k];
#pragma unroll
      for (int sg = -1; sg <= 1; sg += 2) {
        float x = (sg * s[k] - lp[k]) * inv;
        if (x < 0) continue;
        int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        float h1 = lp[k1] + x * lv[k1], h2 = lp[k2] + x * lv[k2];
        if (fabsf(h1) <= s[k1] && fabsf(h2) <= s[k2] && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  return -1;
}

__device__ __forceinline__ F3 geom_normalf(int type, F3 gp, const float* gm, F3 gs, F3 hit) {
  F3 rel = hit - gp;
  if (type == GEOM_PLANE) return colf(gm, 2);
  if (type == GEOM_SPHERE) return normalizedf(rel);
  if (type == GEOM_CAPSULE) {
    F3 axis = colf(gm, 2);
    float h = fminf(fmaxf(dotf(rel, axis), -gs.y), gs.y);
    return normalizedf(rel - axis * h);
  }
  F3 loc = mulTf(gm, rel);
  float ax = fabsf(loc.x) / gs.x, ay = fabsf(loc.y) / gs.y, az = fabsf(loc.z) / gs.z;
  int face = 0;
  float best = ax;
  if (ay > best) { best = ay; face = 1; }
  if (az > best) { best = az; face = 2; }
  float l = face == 0 ? loc.x : (face == 1 ? loc.y : loc.z);
  F3 n = colf(gm, face);
  return l >= 0 ? n : n * -1.0f;
}

}  // namespace mj

#endif
