// Single-precision ray casts for the camera kernel (mjrl_render_kernel): the same intersection routines as the
// rangefinder's fp64 ones in mjrl_collide.h (ray_geom, geom_normal), in float.  An image is 8-bit: a ray that grazes a
// silhouette may fall on the other side of it than the fp64 oracle's ray caster says (tests allow 0.2 % of the pixels),
// nothing else changes -- and fp32 vector arithmetic runs at twice the fp64 rate with half the registers.
#ifndef MJRL_RAYF_H
#define MJRL_RAYF_H

#include "mjrl_collide.h"

namespace mj {

// Reciprocal, square root and reciprocal square root as the hardware's one-instruction approximations (v_rcp_f32,
// v_sqrt_f32, v_rsq_f32: 1 ulp).  The correctly rounded forms the compiler emits for `/` and sqrtf are 10 and 12
// instructions each -- 25 divisions and 19 square roots made a fifth of the ray kernel's instructions -- and the last
// bit of a ray parameter decides nothing an 8-bit image shows beyond the silhouette pixels the tests already allow.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 ldf3(const float* p) { return f3(p[0], p[1], p[2]); }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dotf(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 normalizedf(F3 a) {
  float n2 = dotf(a, a);
  return n2 < 1e-40f ? f3(1, 0, 0) : a * frsq(n2);
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
  float sq = fsqrt(det);
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
    float x = -dotf(rel, n) * frcp(denom);
    if (x < 0) return -1;
    F3 hit = rel + vec * x;
    if (gs.x > 0 && fabsf(dotf(hit, colf(gm, 0))) > gs.x) return -1;
    if (gs.y > 0 && fabsf(dotf(hit, colf(gm, 1))) > gs.y) return -1;
    return x;
  }
  if (type == GEOM_SPHERE) return ray_sphere_atf(gp, gs.x, pnt, vec);
  if (type == GEOM_CAPSULE) {
    // The rangefinder's construction (ray_geom: the side's quadratic, then a sphere at either end, the smallest x >= 0
    // that lies on the capsule), with the three quadratics written in the four products they share -- vec . axis,
    // rel . axis, vec . rel, rel . rel (|vec| = 1) -- instead of through the projected vectors and the caps' own centres:
    // a capsule is what most candidates of a block are (the agents' legs), and this is a third of its instructions.
    const F3 axis = colf(gm, 2);
    const float r = gs.x, len = gs.y;
    float best = -1;
    const float va = dotf(vec, axis), ra = dotf(rel, axis), bv = dotf(vec, rel), rr = dotf(rel, rel);
    const float a = dotf(vec, vec) - va * va, b = bv - va * ra, cc = rr - ra * ra - r * r;
    if (a > 1e-15f) {
      const float det = b * b - a * cc;
      if (det >= 0) {
        const float sq = fsqrt(det), inv = frcp(a);
        const float xs[2] = {(-b - sq) * inv, (-b + sq) * inv};
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const float x = xs[k];
          if (x >= 0 && fabsf(ra + x * va) <= len && (best < 0 || x < best)) best = x;
        }
      }
    }
    const float cap = rr + len * len - r * r, lva = len * va, lra2 = 2.0f * len * ra;
#pragma unroll
    for (int s = -1; s <= 1; s += 2) {
      // sphere of radius r at gp + s len axis: rel_s = rel - s len axis
      const float bs = bv - s * lva, cs = cap - s * lra2, det = bs * bs - cs;
      if (det >= 0) {
        const float sq = fsqrt(det), x0 = -bs - sq, x1 = -bs + sq;
        const float x = x0 >= 0 ? x0 : x1;
        if (x >= 0 && s * (ra + x * va) >= len && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  if (type == GEOM_BOX) {
    // Slab form of the rangefinder's face enumeration (ray_geom: the smallest x >= 0 among the six face planes' hits that
    // lie inside the face): the ray is inside the box for x in [max_k near_k, min_k far_k], so the answer is that
    // interval's start if it is >= 0, else its end -- the same two candidates, a third of the instructions.  An axis the
    // ray runs parallel to constrains nothing if the origin lies inside its slab and everything if not.
    F3 lpv = mulTf(gm, rel), lvv = mulTf(gm, vec);
    const float lp[3] = {lpv.x, lpv.y, lpv.z}, lv[3] = {lvv.x, lvv.y, lvv.z}, s[3] = {gs.x, gs.y, gs.z};
    float tn = -3.0e38f, tf = 3.0e38f;
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const bool par = fabsf(lv[k]) < 1e-15f;
      const float inv = frcp(par ? 1.0f : lv[k]);
      const float t1 = (-s[k] - lp[k]) * inv, t2 = (s[k] - lp[k]) * inv;
      miss = miss || (par && fabsf(lp[k]) > s[k]);
      tn = par ? tn : fmaxf(tn, fminf(t1, t2));
      tf = par ? tf : fminf(tf, fmaxf(t1, t2));
    }
    const float x = tn >= 0 ? tn : tf;
    return (!miss && tn <= tf && x >= 0) ? x : -1;
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
  float ax = fabsf(loc.x) * frcp(gs.x), ay = fabsf(loc.y) * frcp(gs.y), az = fabsf(loc.z) * frcp(gs.z);
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
