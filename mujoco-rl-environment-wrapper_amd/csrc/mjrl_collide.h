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
  if (t1 == GEOM_BOX && t2 == GEOM_BOX) return 16;      // 13 candidates (box_box_item), padded to a power of two
  return 0;
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


// Box against box, one candidate per work item (the same construction, operation for operation, as the oracle's
// ora_box_box_item): separating-axis test over the 6 face normals and 9 edge cross products; on a face axis the incident
// face clipped against the reference face's rectangle -- items 0..7: entry / exit point of each incident edge, items
// 8..11: corners of the reference face strictly inside the incident face's projection -- and on an edge axis item 12, the
// closest points of the two supporting edges.  Every lane of a pair repeats the axis test; nothing is indexed at run time
// (a run-time index would put the axes in scratch memory).  Not inlined: the routine is rare (a movable box near another
// box) and must not shape the register allocation of the step kernel.
__device__ __forceinline__ V3 pick3(int i, V3 a, V3 b, V3 c) { return i == 0 ? a : (i == 1 ? b : c); }
__device__ __forceinline__ real pickr(int i, real a, real b, real c) { return i == 0 ? a : (i == 1 ? b : c); }

#if defined(__HIPCC__)
__device__ __attribute__((noinline))
#else
static inline
#endif
bool box_box_item(int k, real p1x, real p1y, real p1z, const real* m1v, real s1x, real s1y, real s1z, real p2x, real p2y,
                  real p2z, const real* m2v, real s2x, real s2y, real s2z, real margin, real* out) {
  const V3 p1 = v3(p1x, p1y, p1z), p2 = v3(p2x, p2y, p2z);
  V3 A[3], B[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { A[i] = v3(m1v[i], m1v[3 + i], m1v[6 + i]); B[i] = v3(m2v[i], m2v[3 + i], m2v[6 + i]); }
  const real s1[3] = {s1x, s1y, s1z}, s2[3] = {s2x, s2y, s2z};
  const V3 d = p2 - p1;
  real dA[3], dB[3], C[3][3], aC[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++) { dA[i] = dot(d, A[i]); dB[i] = dot(d, B[i]); }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { C[i][j] = dot(A[i], B[j]); aC[i][j] = fabs(C[i][j]); }
  real best = -1e300;
  int code = -1;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    real rb = s2[0] * aC[i][0] + s2[1] * aC[i][1] + s2[2] * aC[i][2];
    real sep = fabs(dA[i]) - (s1[i] + rb);
    if (sep > best) { best = sep; code = i; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    real ra = s1[0] * aC[0][j] + s1[1] * aC[1][j] + s1[2] * aC[2][j];
    real sep = fabs(dB[j]) - (ra + s2[j]);
    if (sep > best) { best = sep; code = 3 + j; }
  }
  if (best > margin) return false;
  real ebest = -1e300, einv = 0;
  int ecode = -1;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      V3 L = cross(A[i], B[j]);
      real l2 = dot(L, L);
      if (l2 < 1e-10) continue;
      real inv = 1.0 / sqrt(l2);
      constexpr int nx[3] = {1, 2, 0}, nn[3] = {2, 0, 1};
      real ra = (s1[nx[i]] * aC[nn[i]][j] + s1[nn[i]] * aC[nx[i]][j]) * inv;
      real rb = (s2[nx[j]] * aC[i][nn[j]] + s2[nn[j]] * aC[i][nx[j]]) * inv;
      real sep = fabs(dot(d, L) * inv) - (ra + rb);
      if (sep > ebest) { ebest = sep; ecode = 3 * i + j; einv = inv; }
    }
  if (ecode >= 0 && ebest > margin) return false;
  const bool use_edge = ecode >= 0 && ebest > best + 0.05 * fabs(best) + 1e-9;
  real dist;
  V3 pos, nrm;
  if (use_edge) {
    if (k != 12) return false;
    const int i = ecode / 3, j = ecode % 3;
    const V3 Ai = pick3(i, A[0], A[1], A[2]), Bj = pick3(j, B[0], B[1], B[2]);
    V3 L = cross(Ai, Bj) * einv;
    if (dot(d, L) < 0) L = L * -1.0;
    V3 e1 = p1, e2 = p2;
#pragma unroll
    for (int q = 0; q < 3; q++) {
      if (q != i) e1 = e1 + A[q] * ((dot(A[q], L) > 0 ? 1.0 : -1.0) * s1[q]);
      if (q != j) e2 = e2 + B[q] * ((dot(B[q], L) > 0 ? -1.0 : 1.0) * s2[q]);
    }
    const V3 w = e1 - e2;
    const real b = pickr(j, pickr(i, C[0][0], C[1][0], C[2][0]), pickr(i, C[0][1], C[1][1], C[2][1]), pickr(i, C[0][2], C[1][2], C[2][2]));
    const real dd = dot(Ai, w), ee = dot(Bj, w), den = 1.0 - b * b;
    real ta = (b * ee - dd) / den, tb = (ee - b * dd) / den;
    const real h1 = pickr(i, s1[0], s1[1], s1[2]), h2 = pickr(j, s2[0], s2[1], s2[2]);
    if (ta > h1) ta = h1;
    if (ta < -h1) ta = -h1;
    if (tb > h2) tb = h2;
    if (tb < -h2) tb = -h2;
    const V3 c1 = e1 + Ai * ta, c2 = e2 + Bj * tb;
    dist = dot(c2 - c1, L);
    if (dist > margin) return false;
    nrm = L;
    pos = (c1 + c2) * 0.5;
  } else {
    if (k >= 12) return false;
    const bool ref1 = code < 3;
    const int ri = ref1 ? code : code - 3;
    const V3 R0 = ref1 ? A[0] : B[0], R1 = ref1 ? A[1] : B[1], R2 = ref1 ? A[2] : B[2];
    const V3 I0 = ref1 ? B[0] : A[0], I1 = ref1 ? B[1] : A[1], I2 = ref1 ? B[2] : A[2];
    const V3 pR = ref1 ? p1 : p2, pI = ref1 ? p2 : p1;
    const real sR0 = ref1 ? s1[0] : s2[0], sR1 = ref1 ? s1[1] : s2[1], sR2 = ref1 ? s1[2] : s2[2];
    const real sI0 = ref1 ? s2[0] : s1[0], sI1 = ref1 ? s2[1] : s1[1], sI2 = ref1 ? s2[2] : s1[2];
    const real dsel = ref1 ? pickr(ri, dA[0], dA[1], dA[2]) : pickr(ri, dB[0], dB[1], dB[2]);
    const real sgn = ref1 ? (dsel >= 0 ? 1.0 : -1.0) : (dsel >= 0 ? -1.0 : 1.0);
    const V3 n = pick3(ri, R0, R1, R2) * sgn;
    const real cj0 = dot(I0, n), cj1 = dot(I1, n), cj2 = dot(I2, n);
    int ii = 0;
    real cmax = cj0;
    if (fabs(cj1) > fabs(cmax)) { ii = 1; cmax = cj1; }
    if (fabs(cj2) > fabs(cmax)) { ii = 2; cmax = cj2; }
    const real msgn = cmax > 0 ? -1.0 : 1.0;
    const int iu = ii == 2 ? 0 : ii + 1, iv = ii == 0 ? 2 : ii - 1;       // (ii+1)%3, (ii+2)%3
    const int ru = ri == 2 ? 0 : ri + 1, rv = ri == 0 ? 2 : ri - 1;
    const V3 Iii = pick3(ii, I0, I1, I2);
    const V3 cI = pI + Iii * (msgn * pickr(ii, sI0, sI1, sI2));
    const V3 cR = pR + n * pickr(ri, sR0, sR1, sR2);
    const V3 eu = pick3(iu, I0, I1, I2) * pickr(iu, sI0, sI1, sI2), ev = pick3(iv, I0, I1, I2) * pickr(iv, sI0, sI1, sI2);
    V3 V[4];
    real X[4], Y[4];
    const V3 tu = pick3(ru, R0, R1, R2), tv = pick3(rv, R0, R1, R2);
    const real hu = pickr(ru, sR0, sR1, sR2), hv = pickr(rv, sR0, sR1, sR2);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const real su = (q == 0 || q == 3) ? 1.0 : -1.0, sv = q < 2 ? 1.0 : -1.0;
      V[q] = (cI + eu * su) + ev * sv;
      const V3 rel = V[q] - cR;
      X[q] = dot(rel, tu); Y[q] = dot(rel, tv);
    }
    const real nout = ref1 ? 1.0 : -1.0;
    if (k < 8) {
      const int e = k >> 1, ex = k & 1;
      // (selects over the four corners: e is the lane's own)
      const V3 Ve = e == 0 ? V[0] : (e == 1 ? V[1] : (e == 2 ? V[2] : V[3]));
      const V3 Vf = e == 0 ? V[1] : (e == 1 ? V[2] : (e == 2 ? V[3] : V[0]));
      const real xa = e == 0 ? X[0] : (e == 1 ? X[1] : (e == 2 ? X[2] : X[3])), ya = e == 0 ? Y[0] : (e == 1 ? Y[1] : (e == 2 ? Y[2] : Y[3]));
      const real xb = e == 0 ? X[1] : (e == 1 ? X[2] : (e == 2 ? X[3] : X[0])), yb = e == 0 ? Y[1] : (e == 1 ? Y[2] : (e == 2 ? Y[3] : Y[0]));
      const real dx = xb - xa, dy = yb - ya;
      real t0 = 0.0, t1 = 1.0;
      bool ok = true;
      const real pp[4] = {-dx, dx, -dy, dy}, qq[4] = {xa + hu, hu - xa, ya + hv, hv - ya};
#pragma unroll
      for (int bnd = 0; bnd < 4; bnd++) {
        if (pp[bnd] == 0.0) { if (qq[bnd] < 0.0) ok = false; continue; }
        const real r = qq[bnd] / pp[bnd];
        if (pp[bnd] < 0.0) { if (r > t0) t0 = r; } else { if (r < t1) t1 = r; }
      }
      if (!ok || t0 > t1) return false;
      if (ex && !(t1 < 1.0)) return false;
      const real t = ex ? t1 : t0;
      const V3 P = Ve + (Vf - Ve) * t;
      const real depth = dot(P - cR, n);
      if (depth > margin) return false;
      dist = depth;
      nrm = n * nout;
      pos = P + n * (-0.5 * depth);
    } else {
      const int q = k - 8;
      const real x = ((q == 0 || q == 3) ? 1.0 : -1.0) * hu, y = (q < 2 ? 1.0 : -1.0) * hv;
      int npos = 0, nneg = 0;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int f = (e + 1) & 3;
        const real w = (X[f] - X[e]) * (y - Y[e]) - (Y[f] - Y[e]) * (x - X[e]);
        npos += w > 0.0; nneg += w < 0.0;
      }
      if (npos != 4 && nneg != 4) return false;
      const V3 mI = Iii * msgn;
      const real nm = dot(n, mI);
      if (nm > -1e-9) return false;
      const V3 Vr = (cR + tu * x) + tv * y;
      const real lam = dot(cI - Vr, mI) / nm;
      if (lam > margin) return false;
      dist = lam;
      nrm = n * nout;
      pos = Vr + n * (0.5 * lam);
    }
  }
  out[0] = dist; out[1] = pos.x; out[2] = pos.y; out[3] = pos.z; out[4] = nrm.x; out[5] = nrm.y; out[6] = nrm.z;
  return true;
}

// evaluate work item k of a (type-ordered) geom pair
__device__ __forceinline__ bool collide_item(int t1, int t2, V3 p1, const M3& m1, V3 s1, V3 p2, const M3& m2, V3 s2,
                                             real margin, int k, RawCon& c, bool boxbox = true) {
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
  if (boxbox && t1 == GEOM_BOX && t2 == GEOM_BOX) {
    real out[7];
    if (!box_box_item(k, p1.x, p1.y, p1.z, m1.m, s1.x, s1.y, s1.z, p2.x, p2.y, p2.z, m2.m, s2.x, s2.y, s2.z, margin, out)) return false;
    c.dist = out[0]; c.pos = v3(out[1], out[2], out[3]); c.n = v3(out[4], out[5], out[6]);
    return true;
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
