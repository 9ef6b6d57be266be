// Narrow phase as independent work items: (candidate pair, item k) -> at most one contact.
// One lane evaluates one item, so a wave covers 64/KMAX candidate pairs per pass and the contacts come
// out already ordered by (pair, k) for the ballot compaction.  Conventions: normal from geom1 to
// geom2, dist < 0 is penetration, pos midway between the surfaces, type(geom1) <= type(geom2).
#ifndef MJRL_COLLIDE_H
#define MJRL_COLLIDE_H

#include "mjrl_math.h"

namespace mj {

enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_BOX = 6 };

struct RawCon {
  real dist;
  V3 pos, n, t;   // t: tangent hint (zero when the pair has no preferred direction)
};

__device__ __forceinline__ int pair_items(int t1, int t2) {
  if (t1 == GEOM_PLANE) return t2 == GEOM_SPHERE ? 1 : (t2 == GEOM_CAPSULE ? 2 : (t2 == GEOM_BOX ? 8 : 0));
  if (t1 == GEOM_SPHERE) return (t2 == GEOM_SPHERE || t2 == GEOM_CAPSULE || t2 == GEOM_BOX) ? 1 : 0;
  if (t1 == GEOM_CAPSULE) return t2 == GEOM_CAPSULE ? 4 : (t2 == GEOM_BOX ? 2 : 0);
  return 0;   // box-box: not generated
}

// orthonormal contact frame rows (n, t1, t2) from the normal and an optional tangent hint
__device__ __forceinline__ void make_frame(V3 n, V3 hint, real* frame) {
  n = normalized(n, 0);
  V3 t = hint;
  if (norm(t) < 0.5) t = (n.y < 0.5 && n.y > -0.5) ? v3(0, 1, 0) : v3(0, 0, 1);
  t = t - n * dot(n, t);
  t = normalized(t, 0);
  V3 b = cross(n, t);
  st3(frame, n); st3(frame + 3, t); st3(frame + 6, b);
}

__device__ __forceinline__ bool sphere_sphere(V3 p1, real r1, V3 p2, real r2, real margin, RawCon& c) {
  V3 dif = p2 - p1;
  real cdist = norm(dif);
  if (cdist > margin + r1 + r2) return false;
  c.dist = cdist - r1 - r2;
  c.n = normalized(dif, 0);
  c.t = v3(0, 0, 0);
  c.pos = p1 + c.n * (r1 + 0.5 * c.dist);
  return true;
}

__device__ __forceinline__ bool plane_sphere(V3 pp, V3 pn, V3 sp, real r, real margin, RawCon& c) {
  real cdist = dot(sp - pp, pn);
  if (cdist > margin + r) return false;
  c.dist = cdist - r;
  c.n = pn;
  c.t = v3(0, 0, 0);
  c.pos = sp + pn * (-(r + 0.5 * c.dist));
  return true;
}

// sphere against box; normal points sphere -> box
__device__ __forceinline__ bool sphere_box(V3 sp, real r, V3 bp, const M3& bm, V3 bs, real margin, RawCon& c) {
  V3 loc = mulT(bm, sp - bp);
  real l[3] = {loc.x, loc.y, loc.z}, s[3] = {bs.x, bs.y, bs.z}, cl[3];
  bool inside = true;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    cl[k] = l[k];
    if (cl[k] > s[k]) { cl[k] = s[k]; inside = false; }
    else if (cl[k] < -s[k]) { cl[k] = -s[k]; inside = false; }
  }
  V3 nout;
  real d;
  if (!inside) {
    V3 dif = v3(l[0] - cl[0], l[1] - cl[1], l[2] - cl[2]);
    d = norm(dif);
    if (d - r > margin) return false;
    nout = dif * (1.0 / d);
  } else {
    int best = 0;
    real bestpen = s[0] - fabs(l[0]);
#pragma unroll
    for (int k = 1; k < 3; k++) {
      real pen = s[k] - fabs(l[k]);
      if (pen < bestpen) { bestpen = pen; best = k; }
    }
    // (selects instead of l[best] / cl[best]: a run-time index would put the three arrays in scratch memory)
    real lb = best == 0 ? l[0] : (best == 1 ? l[1] : l[2]), sb = best == 0 ? s[0] : (best == 1 ? s[1] : s[2]);
    real sg = lb >= 0 ? 1.0 : -1.0;
    nout = v3(best == 0 ? sg : 0.0, best == 1 ? sg : 0.0, best == 2 ? sg : 0.0);
    real face = sg * sb;
    cl[0] = best == 0 ? face : cl[0]; cl[1] = best == 1 ? face : cl[1]; cl[2] = best == 2 ? face : cl[2];
    d = -bestpen;
  }
  c.dist = d - r;
  V3 nw = mul(bm, nout);
  V3 cw = mul(bm, v3(cl[0], cl[1], cl[2])) + bp;
  c.n = nw * -1.0;
  c.t = v3(0, 0, 0);
  c.pos = cw + nw * (0.5 * c.dist);
  return true;
}

__device__ __forceinline__ real segbox_slope(V3 p0, V3 ax, V3 bs, real t) {
  real p[3] = {p0.x, p0.y, p0.z}, a[3] = {ax.x, ax.y, ax.z}, s[3] = {bs.x, bs.y, bs.z}, g = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    real x = p[k] + t * a[k], e = 0;
    if (x > s[k]) e = x - s[k]; else if (x < -s[k]) e = x + s[k];
    g += e * a[k];
  }
  return g;
}
__device__ __forceinline__ real pointbox_dist(V3 p0, V3 ax, V3 bs, real t) {
  real p[3] = {p0.x, p0.y, p0.z}, a[3] = {ax.x, ax.y, ax.z}, s[3] = {bs.x, bs.y, bs.z}, acc = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    real x = p[k] + t * a[k], e = 0;
    if (x > s[k]) e = x - s[k]; else if (x < -s[k]) e = x + s[k];
    acc += e * e;
  }
  return sqrt(acc);
}

__device__ __forceinline__ bool capsule_box(V3 cp, const M3& cm, V3 cs, V3 bp, const M3& bm, V3 bs, real margin, int k,
                                            RawCon& c) {
  V3 axis = col(cm, 2);
  real len = cs.y, r = cs.x;
  V3 p0 = mulT(bm, cp - bp), ax = mulT(bm, axis);
  real lo = -len, hi = len, tstar;
  real glo = segbox_slope(p0, ax, bs, lo), ghi = segbox_slope(p0, ax, bs, hi);
  if (glo >= 0) tstar = lo;
  else if (ghi <= 0) tstar = hi;
  else {
    // the slope is piecewise linear in t with kinks where a coordinate crosses a box face: bracket the root between
    // kinks, then solve the linear piece exactly
    real pk[3] = {p0.x, p0.y, p0.z}, ak[3] = {ax.x, ax.y, ax.z}, sk[3] = {bs.x, bs.y, bs.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (fabs(ak[k]) < MJ_MINVAL) continue;
#pragma unroll
      for (int s = -1; s <= 1; s += 2) {
        real tb = (s * sk[k] - pk[k]) / ak[k];
        if (tb <= lo || tb >= hi) continue;
        real gb = segbox_slope(p0, ax, bs, tb);
        if (gb < 0) { lo = tb; glo = gb; } else { hi = tb; ghi = gb; }
      }
    }
    tstar = lo - glo * (hi - lo) / (ghi - glo);
  }
  real dstar = pointbox_dist(p0, ax, bs, tstar);
  real dpos = pointbox_dist(p0, ax, bs, len), dneg = pointbox_dist(p0, ax, bs, -len);
  real dend = dpos < dneg ? dpos : dneg;
  bool interior = (tstar > -len && tstar < len && dstar < dend - 1e-9);
  real t;
  if (interior) t = (k == 0) ? tstar : (dpos <= dneg ? len : -len);
  else t = (k == 0) ? len : -len;
  bool hit = sphere_box(cp + axis * t, r, bp, bm, bs, margin, c);
  c.t = axis;
  return hit;
}

__device__ __forceinline__ bool capsule_capsule(V3 p1, const M3& m1, V3 s1, V3 p2, const M3& m2, V3 s2, real margin, int k,
                                                RawCon& c) {
  V3 a1 = col(m1, 2), a2 = col(m2, 2), dif = p1 - p2;
  real len1 = s1.y, len2 = s2.y;
  real mb = -dot(a1, a2), u = -dot(a1, dif), v = dot(a2, dif);
  real det = 1.0 - mb * mb;
  real x1, x2;
  if (fabs(det) >= 1e-12) {
    if (k != 0) return false;
    x1 = (u - mb * v) / det; x2 = (v - mb * u) / det;
    if (x1 > len1) { x1 = len1; x2 = v - mb * len1; }
    else if (x1 < -len1) { x1 = -len1; x2 = v + mb * len1; }
    if (x2 > len2) {
      x2 = len2; x1 = u - mb * len2;
      if (x1 > len1) x1 = len1; else if (x1 < -len1) x1 = -len1;
    } else if (x2 < -len2) {
      x2 = -len2; x1 = u + mb * len2;
      if (x1 > len1) x1 = len1; else if (x1 < -len1) x1 = -len1;
    }
  } else if (k < 2) {
    x1 = (k == 0) ? len1 : -len1;
    x2 = v - mb * x1;
    if (x2 > len2 || x2 < -len2) return false;
  } else {
    x2 = (k == 2) ? len2 : -len2;
    x1 = u - mb * x2;
    if (x1 >= len1 || x1 <= -len1) return false;
  }
  return sphere_sphere(p1 + a1 * x1, s1.x, p2 + a2 * x2, s2.x, margin, c);
}

// evaluate work item k of a (type-ordered) geom pair
__device__ __forceinline__ bool collide_item(int t1, int t2, V3 p1, const M3& m1, V3 s1, V3 p2, const M3& m2, V3 s2,
                                             real margin, int k, RawCon& c) {
  // (defined on every path: with the record undefined where no routine writes it, the compiler kept it in scratch
  // memory instead of registers)
  c.dist = 0; c.pos = v3(0, 0, 0); c.n = v3(0, 0, 0); c.t = v3(0, 0, 0);
  if (t1 == GEOM_PLANE) {
    V3 pn = col(m1, 2);
    if (t2 == GEOM_SPHERE) return plane_sphere(p1, pn, p2, s2.x, margin, c);
    if (t2 == GEOM_CAPSULE) {
      V3 axis = col(m2, 2);
      bool hit = plane_sphere(p1, pn, p2 + axis * (k == 0 ? s2.y : -s2.y), s2.x, margin, c);
      c.t = axis;
      return hit;
    }
    if (t2 == GEOM_BOX) {
      V3 off = mul(m2, v3((k & 1) ? s2.x : -s2.x, (k & 2) ? s2.y : -s2.y, (k & 4) ? s2.z : -s2.z));
      real ldist = dot(p2 - p1, pn) + dot(off, pn);
      if (ldist > margin) return false;
      c.dist = ldist;
      c.n = pn;
      c.t = v3(0, 0, 0);
      c.pos = p2 + off + pn * (-0.5 * ldist);
      return true;
    }
    return false;
  }
  if (t1 == GEOM_SPHERE) {
    if (t2 == GEOM_SPHERE) return sphere_sphere(p1, s1.x, p2, s2.x, margin, c);
    if (t2 == GEOM_CAPSULE) {
      V3 axis = col(m2, 2);
      real x = fmin(fmax(dot(axis, p1 - p2), -s2.y), s2.y);
      return sphere_sphere(p1, s1.x, p2 + axis * x, s2.x, margin, c);
    }
    if (t2 == GEOM_BOX) return sphere_box(p1, s1.x, p2, m2, s2, margin, c);
    return false;
  }
  if (t1 == GEOM_CAPSULE) {
    if (t2 == GEOM_CAPSULE) return capsule_capsule(p1, m1, s1, p2, m2, s2, margin, k, c);
    if (t2 == GEOM_BOX) return capsule_box(p1, m1, s1, p2, m2, s2, margin, k, c);
  }
  return false;
}

// ---- ray casts for the rangefinder: distance along the unit ray or -1
__device__ __forceinline__ real ray_sphere_at(V3 center, real r, V3 pnt, V3 vec) {
  V3 rel = pnt - center;
  real b = dot(vec, rel), cc = dot(rel, rel) - r * r;
  real det = b * b - cc;
  if (det < 0) return -1;
  real sq = sqrt(det);
  real x0 = -b - sq, x1 = -b + sq;
  if (x0 >= 0) return x0;
  if (x1 >= 0) return x1;
  return -1;
}
__device__ __forceinline__ real ray_geom(int type, V3 gp, const M3& gm, V3 gs, V3 pnt, V3 vec) {
  V3 rel = pnt - gp;
  if (type == GEOM_PLANE) {
    V3 n = col(gm, 2);
    real denom = dot(vec, n);
    if (denom > -MJ_MINVAL) return -1;
    real x = -dot(rel, n) / denom;
    if (x < 0) return -1;
    V3 hit = rel + vec * x;
    if (gs.x > 0 && fabs(dot(hit, col(gm, 0))) > gs.x) return -1;
    if (gs.y > 0 && fabs(dot(hit, col(gm, 1))) > gs.y) return -1;
    return x;
  }
  if (type == GEOM_SPHERE) return ray_sphere_at(gp, gs.x, pnt, vec);
  if (type == GEOM_CAPSULE) {
    V3 axis = col(gm, 2);
    real r = gs.x, len = gs.y, best = -1;
    real va = dot(vec, axis), ra = dot(rel, axis);
    V3 vp = vec - axis * va, rp = rel - axis * ra;
    real a = dot(vp, vp), b = dot(vp, rp), cc = dot(rp, rp) - r * r;
    if (a > MJ_MINVAL) {
      real det = b * b - a * cc;
      if (det >= 0) {
        real sq = sqrt(det);
        real xs[2] = {(-b - sq) / a, (-b + sq) / a};
#pragma unroll
        for (int k = 0; k < 2; k++) {
          real x = xs[k];
          if (x >= 0 && fabs(ra + x * va) <= len && (best < 0 || x < best)) best = x;
        }
      }
    }
#pragma unroll
    for (int s = -1; s <= 1; s += 2) {
      real x = ray_sphere_at(gp + axis * (s * len), r, pnt, vec);
      if (x >= 0) {
        real h = ra + x * va;
        if (s * h >= len && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  if (type == GEOM_BOX) {
    V3 lpv = mulT(gm, rel), lvv = mulT(gm, vec);
    real lp[3] = {lpv.x, lpv.y, lpv.z}, lv[3] = {lvv.x, lvv.y, lvv.z}, s[3] = {gs.x, gs.y, gs.z};
    real best = -1;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (fabs(lv[k]) < MJ_MINVAL) continue;
#pragma unroll
      for (int sg = -1; sg <= 1; sg += 2) {
        real x = (sg * s[k] - lp[k]) / lv[k];
        if (x < 0) continue;
        int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
        real h1 = lp[k1] + x * lv[k1], h2 = lp[k2] + x * lv[k2];
        if (fabs(h1) <= s[k1] && fabs(h2) <= s[k2] && (best < 0 || x < best)) best = x;
      }
    }
    return best;
  }
  return -1;
}

// outward surface normal of a geom at a point on its surface (shading of the ray-cast camera)
__device__ __forceinline__ V3 geom_normal(int type, V3 gp, const M3& gm, V3 gs, V3 hit) {
  V3 rel = hit - gp;
  if (type == GEOM_PLANE) return col(gm, 2);
  if (type == GEOM_SPHERE) return normalized(rel, 0);
  if (type == GEOM_CAPSULE) {
    V3 axis = col(gm, 2);
    real h = fmin(fmax(dot(rel, axis), -gs.y), gs.y);
    return normalized(rel - axis * h, 0);
  }
  V3 loc = mulT(gm, rel);
  real ax = fabs(loc.x) / gs.x, ay = fabs(loc.y) / gs.y, az = fabs(loc.z) / gs.z;
  int face = 0;
  real best = ax;
  if (ay > best) { best = ay; face = 1; }
  if (az > best) { best = az; face = 2; }
  real l = face == 0 ? loc.x : (face == 1 ? loc.y : loc.z);
  V3 n = col(gm, face);
  return l >= 0 ? n : n * -1.0;
}

}  // namespace mj

#endif
