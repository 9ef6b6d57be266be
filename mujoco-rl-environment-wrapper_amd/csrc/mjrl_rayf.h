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

// ------------------------------------------------------------------ two rays per lane
// The same routines for TWO rays at once, the pair in the halves of 64-bit registers: multiplies, adds and fused
// multiply-adds of a pair are one packed instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, the full-rate fp32 of
// this chip); comparisons, selects and the transcendental unit's rcp / sqrt / rsq stay one instruction per ray.  Every
// expression is the scalar routine's, term for term (the same products feed the same sums, so the front end contracts
// the same multiply-adds): a pair's results are the two scalar results.  Where the scalar routine returns early the
// pair computes on and selects, with arguments clamped where the discarded branch would leave the domain.
typedef float f2 __attribute__((ext_vector_type(2)));
struct B2 { bool x, y; };                                    // a condition per ray
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 splat2(float a) { return mk2(a, a); }
__device__ __forceinline__ f2 sel2(B2 c, f2 a, f2 b) { return mk2(c.x ? a.x : b.x, c.y ? a.y : b.y); }
__device__ __forceinline__ B2 and2(B2 a, B2 b) { B2 r; r.x = a.x && b.x; r.y = a.y && b.y; return r; }
__device__ __forceinline__ B2 or2(B2 a, B2 b) { B2 r; r.x = a.x || b.x; r.y = a.y || b.y; return r; }
__device__ __forceinline__ B2 not2(B2 a) { B2 r; r.x = !a.x; r.y = !a.y; return r; }
__device__ __forceinline__ B2 lt2(f2 a, f2 b) { B2 r; r.x = a.x < b.x; r.y = a.y < b.y; return r; }
__device__ __forceinline__ B2 le2(f2 a, f2 b) { B2 r; r.x = a.x <= b.x; r.y = a.y <= b.y; return r; }
__device__ __forceinline__ B2 gt2(f2 a, f2 b) { B2 r; r.x = a.x > b.x; r.y = a.y > b.y; return r; }
__device__ __forceinline__ B2 ge2(f2 a, f2 b) { B2 r; r.x = a.x >= b.x; r.y = a.y >= b.y; return r; }
__device__ __forceinline__ f2 frcp2(f2 a) { return mk2(frcp(a.x), frcp(a.y)); }
__device__ __forceinline__ f2 fsqrt2(f2 a) { return mk2(fsqrt(a.x), fsqrt(a.y)); }
__device__ __forceinline__ f2 frsq2(f2 a) { return mk2(frsq(a.x), frsq(a.y)); }
__device__ __forceinline__ f2 fabs2(f2 a) { return mk2(fabsf(a.x), fabsf(a.y)); }
__device__ __forceinline__ f2 fmin2(f2 a, f2 b) { return mk2(fminf(a.x, b.x), fminf(a.y, b.y)); }
__device__ __forceinline__ f2 fmax2(f2 a, f2 b) { return mk2(fmaxf(a.x, b.x), fmaxf(a.y, b.y)); }

struct P3 { f2 x, y, z; };                                   // a 3-vector per ray
__device__ __forceinline__ P3 p3(f2 x, f2 y, f2 z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ P3 splat3(F3 a) { return p3(splat2(a.x), splat2(a.y), splat2(a.z)); }
__device__ __forceinline__ P3 pair3(F3 a, F3 b) { return p3(mk2(a.x, b.x), mk2(a.y, b.y), mk2(a.z, b.z)); }
__device__ __forceinline__ F3 first3(P3 a) { return f3(a.x.x, a.y.x, a.z.x); }
__device__ __forceinline__ F3 second3(P3 a) { return f3(a.x.y, a.y.y, a.z.y); }
__device__ __forceinline__ P3 operator+(P3 a, P3 b) { return p3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ P3 operator-(P3 a, P3 b) { return p3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ P3 operator*(P3 a, f2 s) { return p3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f2 dot2(P3 a, P3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ P3 normalized2(P3 a) {
  const f2 n2 = dot2(a, a);
  const B2 tiny = lt2(n2, splat2(1e-40f));
  const f2 s = frsq2(fmax2(n2, splat2(1e-40f)));
  return p3(sel2(tiny, splat2(1.0f), a.x * s), sel2(tiny, splat2(0.0f), a.y * s), sel2(tiny, splat2(0.0f), a.z * s));
}
__device__ __forceinline__ P3 mul2(const float* m, P3 v) {
  return p3(splat2(m[0]) * v.x + splat2(m[1]) * v.y + splat2(m[2]) * v.z, splat2(m[3]) * v.x + splat2(m[4]) * v.y + splat2(m[5]) * v.z,
            splat2(m[6]) * v.x + splat2(m[7]) * v.y + splat2(m[8]) * v.z);
}
__device__ __forceinline__ P3 mulT2(const float* m, P3 v) {
  return p3(splat2(m[0]) * v.x + splat2(m[3]) * v.y + splat2(m[6]) * v.z, splat2(m[1]) * v.x + splat2(m[4]) * v.y + splat2(m[7]) * v.z,
            splat2(m[2]) * v.x + splat2(m[5]) * v.y + splat2(m[8]) * v.z);
}

// ray_geomf for a pair of rays against ONE geom (type, frame and size are the pair's, the rays' origins and directions
// their own); `on`: the ray takes part at all (its result is -1 otherwise)
__device__ __forceinline__ f2 ray_geom2(int type, B2 on, F3 gp, const float* gm, F3 gs, P3 pnt, P3 vec) {
  const f2 none = splat2(-1.0f), zero = splat2(0.0f);
  const P3 rel = pnt - splat3(gp);
  f2 res = none;
  if (type == GEOM_PLANE) {
    const P3 n = splat3(colf(gm, 2));
    const f2 denom = dot2(vec, n);
    const B2 front = not2(gt2(denom, splat2(-1e-15f)));
    const f2 x = -dot2(rel, n) * frcp2(sel2(front, denom, splat2(-1.0f)));
    const P3 hit = rel + vec * x;
    B2 ok = and2(front, not2(lt2(x, zero)));
    if (gs.x > 0) ok = and2(ok, not2(gt2(fabs2(dot2(hit, splat3(colf(gm, 0)))), splat2(gs.x))));
    if (gs.y > 0) ok = and2(ok, not2(gt2(fabs2(dot2(hit, splat3(colf(gm, 1)))), splat2(gs.y))));
    res = sel2(ok, x, none);
  } else if (type == GEOM_SPHERE) {
    const f2 b = dot2(vec, rel), cc = dot2(rel, rel) - splat2(gs.x) * splat2(gs.x);
    const f2 det = b * b - cc;
    const f2 sq = fsqrt2(fmax2(det, zero));
    const f2 x0 = -b - sq, x1 = -b + sq;
    res = sel2(lt2(det, zero), none, sel2(ge2(x0, zero), x0, sel2(ge2(x1, zero), x1, none)));
  } else if (type == GEOM_CAPSULE) {
    const P3 axis = splat3(colf(gm, 2));
    const float r = gs.x, len = gs.y;
    f2 best = none;
    const f2 va = dot2(vec, axis), ra = dot2(rel, axis), bv = dot2(vec, rel), rr = dot2(rel, rel);
    const f2 a = dot2(vec, vec) - va * va, b = bv - va * ra, cc = rr - ra * ra - splat2(r) * splat2(r);
    {
      const f2 det = b * b - a * cc;
      const B2 side = and2(gt2(a, splat2(1e-15f)), ge2(det, zero));
      const f2 sq = fsqrt2(fmax2(det, zero)), inv = frcp2(sel2(gt2(a, splat2(1e-15f)), a, splat2(1.0f)));
      const f2 xs[2] = {(-b - sq) * inv, (-b + sq) * inv};
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const f2 x = xs[k];
        const B2 take = and2(and2(side, ge2(x, zero)), and2(le2(fabs2(ra + x * va), splat2(len)), or2(lt2(best, zero), lt2(x, best))));
        best = sel2(take, x, best);
      }
    }
    const f2 cap = rr + splat2(len) * splat2(len) - splat2(r) * splat2(r), lva = splat2(len) * va, lra2 = splat2(2.0f * len) * ra;
#pragma unroll
    for (int s = -1; s <= 1; s += 2) {
      const f2 sf = splat2((float)s);
      const f2 bs = bv - sf * lva, cs = cap - sf * lra2, det = bs * bs - cs;
      const f2 sq = fsqrt2(fmax2(det, zero)), x0 = -bs - sq, x1 = -bs + sq;
      const f2 x = sel2(ge2(x0, zero), x0, x1);
      const B2 take = and2(and2(ge2(det, zero), ge2(x, zero)), and2(ge2(sf * (ra + x * va), splat2(len)), or2(lt2(best, zero), lt2(x, best))));
      best = sel2(take, x, best);
    }
    res = best;
  } else if (type == GEOM_BOX) {
    const P3 lpv = mulT2(gm, rel), lvv = mulT2(gm, vec);
    const f2 lp[3] = {lpv.x, lpv.y, lpv.z}, lv[3] = {lvv.x, lvv.y, lvv.z};
    const float s[3] = {gs.x, gs.y, gs.z};
    f2 tn = splat2(-3.0e38f), tf = splat2(3.0e38f);
    B2 miss; miss.x = false; miss.y = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const B2 par = lt2(fabs2(lv[k]), splat2(1e-15f));
      const f2 inv = frcp2(sel2(par, splat2(1.0f), lv[k]));
      const f2 t1 = (splat2(-s[k]) - lp[k]) * inv, t2 = (splat2(s[k]) - lp[k]) * inv;
      miss = or2(miss, and2(par, gt2(fabs2(lp[k]), splat2(s[k]))));
      tn = sel2(par, tn, fmax2(tn, fmin2(t1, t2)));
      tf = sel2(par, tf, fmin2(tf, fmax2(t1, t2)));
    }
    const f2 x = sel2(ge2(tn, zero), tn, tf);
    res = sel2(and2(and2(not2(miss), le2(tn, tf)), ge2(x, zero)), x, none);
  }
  return sel2(on, res, none);
}

}  // namespace mj

#endif
