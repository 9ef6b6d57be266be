/* ORACLE -- TEST INFRASTRUCTURE ONLY (see ora_math.h for the provenance statement).
 *
 * Narrow-phase contact generation for the primitive pairs the shipped levels contain
 * (plane, sphere, capsule, box).  Conventions follow MuJoCo's documented contact model:
 * the normal points from geom1 to geom2, dist<0 is penetration, pos is the midpoint between
 * the two surfaces, and the pair is ordered so that type(geom1) <= type(geom2).
 * The capsule-box routine is a from-scratch geometric construction (closest point of the
 * segment to the box, plus end caps), not a restatement of MuJoCo's feature walk.
 */
#ifndef ORA_COLLIDE_H
#define ORA_COLLIDE_H

#include "ora_math.h"

typedef struct {
  double dist;
  double pos[3];
  double frame[9]; /* rows: normal, tangent1, tangent2 (tangent1 may be a hint or zero before make_frame) */
} ora_rawcon;

enum { ORA_GEOM_PLANE = 0, ORA_GEOM_SPHERE = 2, ORA_GEOM_CAPSULE = 3, ORA_GEOM_BOX = 6 };

/* complete an orthonormal frame from frame[0..2] (normal) and an optional hint in frame[3..5] */
static inline void ora_make_frame(double* frame) {
  v3_normalize(frame);
  if (v3_norm(frame + 3) < 0.5) {
    v3_zero(frame + 3);
    if (frame[1] < 0.5 && frame[1] > -0.5) frame[4] = 1.0; else frame[5] = 1.0;
  }
  double d = v3_dot(frame, frame + 3);
  v3_addscl(frame + 3, frame + 3, frame, -d);
  v3_normalize(frame + 3);
  v3_cross(frame + 6, frame, frame + 3);
}

static inline int ora_sphere_sphere_raw(ora_rawcon* c, const double* p1, double r1, const double* p2, double r2,
                                        double margin) {
  double dif[3];
  v3_sub(dif, p2, p1);
  double cdist = v3_norm(dif);
  if (cdist > margin + r1 + r2) return 0;
  c->dist = cdist - r1 - r2;
  v3_copy(c->frame, dif);
  v3_normalize(c->frame);
  v3_zero(c->frame + 3);
  v3_addscl(c->pos, p1, c->frame, r1 + 0.5 * c->dist);
  return 1;
}

static inline int ora_plane_sphere_raw(ora_rawcon* c, const double* ppos, const double* pmat, const double* spos,
                                       double r, double margin) {
  double n[3] = {pmat[2], pmat[5], pmat[8]}, dif[3];
  v3_sub(dif, spos, ppos);
  double cdist = v3_dot(dif, n);
  if (cdist > margin + r) return 0;
  c->dist = cdist - r;
  v3_copy(c->frame, n);
  v3_zero(c->frame + 3);
  v3_addscl(c->pos, spos, n, -(r + 0.5 * c->dist));
  return 1;
}

static inline int ora_plane_capsule(ora_rawcon* c, const double* ppos, const double* pmat, const double* cpos,
                                    const double* cmat, const double* csize, double margin) {
  double axis[3] = {cmat[2], cmat[5], cmat[8]}, end[3];
  int n = 0;
  for (int s = 1; s >= -1; s -= 2) {
    v3_addscl(end, cpos, axis, s * csize[1]);
    if (ora_plane_sphere_raw(c + n, ppos, pmat, end, csize[0], margin)) {
      v3_copy(c[n].frame + 3, axis); /* align the first tangent with the capsule */
      n++;
    }
  }
  return n;
}

static inline int ora_plane_box(ora_rawcon* c, const double* ppos, const double* pmat, const double* bpos,
                                const double* bmat, const double* bsize, double margin) {
  double n[3] = {pmat[2], pmat[5], pmat[8]}, dif[3];
  v3_sub(dif, bpos, ppos);
  double cdist = v3_dot(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {   /* every corner inside the margin makes a contact (nconmax is the only cap) */
    double loc[3] = {(i & 1 ? bsize[0] : -bsize[0]), (i & 2 ? bsize[1] : -bsize[1]), (i & 4 ? bsize[2] : -bsize[2])};
    double off[3];
    m3_mulv(off, bmat, loc);
    double ldist = cdist + v3_dot(off, n);
    if (ldist > margin) continue;
    c[cnt].dist = ldist;
    v3_copy(c[cnt].frame, n);
    v3_zero(c[cnt].frame + 3);
    double corner[3];
    v3_add(corner, bpos, off);
    v3_addscl(c[cnt].pos, corner, n, -0.5 * ldist);
    cnt++;
  }
  return cnt;
}

static inline int ora_sphere_capsule(ora_rawcon* c, const double* spos, double r, const double* cpos,
                                     const double* cmat, const double* csize, double margin) {
  double axis[3] = {cmat[2], cmat[5], cmat[8]}, vec[3], pt[3];
  v3_sub(vec, spos, cpos);
  double x = v3_dot(axis, vec);
  if (x > csize[1]) x = csize[1];
  if (x < -csize[1]) x = -csize[1];
  v3_addscl(pt, cpos, axis, x);
  return ora_sphere_sphere_raw(c, spos, r, pt, csize[0], margin);
}

static inline int ora_capsule_capsule(ora_rawcon* c, const double* p1, const double* m1, const double* s1,
                                      const double* p2, const double* m2, const double* s2, double margin) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]}, dif[3];
  double len1 = s1[1], len2 = s2[1];
  v3_sub(dif, p1, p2);
  double mb = -v3_dot(a1, a2), u = -v3_dot(a1, dif), v = v3_dot(a2, dif);
  double det = 1.0 - mb * mb;
  double e1[3], e2[3];
  if (fabs(det) >= 1e-12) {
    double x1 = (u - mb * v) / det, x2 = (v - mb * u) / det;
    if (x1 > len1) { x1 = len1; x2 = v - mb * len1; }
    else if (x1 < -len1) { x1 = -len1; x2 = v + mb * len1; }
    if (x2 > len2) {
      x2 = len2; x1 = u - mb * len2;
      if (x1 > len1) x1 = len1; else if (x1 < -len1) x1 = -len1;
    } else if (x2 < -len2) {
      x2 = -len2; x1 = u + mb * len2;
      if (x1 > len1) x1 = len1; else if (x1 < -len1) x1 = -len1;
    }
    v3_addscl(e1, p1, a1, x1);
    v3_addscl(e2, p2, a2, x2);
    return ora_sphere_sphere_raw(c, e1, s1[0], e2, s2[0], margin);
  }
  /* parallel axes: pair each end of one segment with its projection on the other (four independent
   * candidates; generically two of them are valid) */
  int n = 0;
  for (int k = 0; k < 4; k++) {
    double x1, x2;
    if (k < 2) {
      x1 = (k == 0) ? len1 : -len1;
      x2 = v - mb * x1;
      if (x2 > len2 || x2 < -len2) continue;
    } else {
      x2 = (k == 2) ? len2 : -len2;
      x1 = u - mb * x2;
      if (x1 >= len1 || x1 <= -len1) continue; /* ends already covered above */
    }
    v3_addscl(e1, p1, a1, x1);
    v3_addscl(e2, p2, a2, x2);
    n += ora_sphere_sphere_raw(c + n, e1, s1[0], e2, s2[0], margin);
  }
  return n;
}

/* sphere (centre spos, radius r) against a box; normal points sphere -> box */
static inline int ora_sphere_box(ora_rawcon* c, const double* spos, double r, const double* bpos,
                                 const double* bmat, const double* bsize, double margin) {
  double rel[3], loc[3], clamped[3], dif[3];
  v3_sub(rel, spos, bpos);
  m3_mulTv(loc, bmat, rel);
  int inside = 1;
  for (int k = 0; k < 3; k++) {
    clamped[k] = loc[k];
    if (clamped[k] > bsize[k]) { clamped[k] = bsize[k]; inside = 0; }
    else if (clamped[k] < -bsize[k]) { clamped[k] = -bsize[k]; inside = 0; }
  }
  double nout[3], d;
  if (!inside) {
    v3_sub(dif, loc, clamped);
    d = v3_norm(dif);
    if (d - r > margin) return 0;
    v3_scl(nout, dif, 1.0 / d);
  } else {
    int best = 0;
    double bestpen = bsize[0] - fabs(loc[0]);
    for (int k = 1; k < 3; k++) {
      double pen = bsize[k] - fabs(loc[k]);
      if (pen < bestpen) { bestpen = pen; best = k; }
    }
    v3_zero(nout);
    nout[best] = loc[best] >= 0 ? 1.0 : -1.0;
    clamped[best] = nout[best] * bsize[best];
    d = -bestpen;
  }
  c->dist = d - r;
  double nw[3], cw[3];
  m3_mulv(nw, bmat, nout);
  m3_mulv(cw, bmat, clamped);
  v3_add(cw, cw, bpos);
  v3_scl(c->frame, nw, -1.0);
  v3_zero(c->frame + 3);
  v3_addscl(c->pos, cw, nw, 0.5 * c->dist);
  return 1;
}

/* derivative of half the squared distance between the box and the point pos + t*axis (box frame) */
static inline double ora_segbox_slope(const double* p0, const double* ax, const double* bsize, double t) {
  double g = 0.0;
  for (int k = 0; k < 3; k++) {
    double x = p0[k] + t * ax[k], e = 0.0;
    if (x > bsize[k]) e = x - bsize[k]; else if (x < -bsize[k]) e = x + bsize[k];
    g += e * ax[k];
  }
  return g;
}
static inline double ora_pointbox_dist(const double* p0, const double* ax, const double* bsize, double t) {
  double s = 0.0;
  for (int k = 0; k < 3; k++) {
    double x = p0[k] + t * ax[k], e = 0.0;
    if (x > bsize[k]) e = x - bsize[k]; else if (x < -bsize[k]) e = x + bsize[k];
    s += e * e;
  }
  return sqrt(s);
}

static inline int ora_capsule_box(ora_rawcon* c, const double* cpos, const double* cmat, const double* csize,
                                  const double* bpos, const double* bmat, const double* bsize, double margin) {
  double axis_w[3] = {cmat[2], cmat[5], cmat[8]}, rel[3], p0[3], ax[3];
  double len = csize[1], r = csize[0];
  v3_sub(rel, cpos, bpos);
  m3_mulTv(p0, bmat, rel);
  m3_mulTv(ax, bmat, axis_w);
  /* closest parameter on the segment: root of a monotone slope that is piecewise linear in t, with kinks where a
   * coordinate crosses a box face.  Bracket the root between kinks, then solve the linear piece exactly. */
  double lo = -len, hi = len, tstar;
  double glo = ora_segbox_slope(p0, ax, bsize, lo), ghi = ora_segbox_slope(p0, ax, bsize, hi);
  if (glo >= 0.0) tstar = lo;
  else if (ghi <= 0.0) tstar = hi;
  else {
    for (int k = 0; k < 3; k++) {
      if (fabs(ax[k]) < ORA_MINVAL) continue;
      for (int s = -1; s <= 1; s += 2) {
        double tb = (s * bsize[k] - p0[k]) / ax[k];
        if (tb <= lo || tb >= hi) continue;
        double gb = ora_segbox_slope(p0, ax, bsize, tb);
        if (gb < 0.0) { lo = tb; glo = gb; } else { hi = tb; ghi = gb; }
      }
    }
    tstar = lo - glo * (hi - lo) / (ghi - glo);
  }
  double dstar = ora_pointbox_dist(p0, ax, bsize, tstar);
  double dpos = ora_pointbox_dist(p0, ax, bsize, len), dneg = ora_pointbox_dist(p0, ax, bsize, -len);
  double dend = dpos < dneg ? dpos : dneg;
  int n = 0;
  double pt[3];
  int interior = (tstar > -len && tstar < len && dstar < dend - 1e-9);
  if (interior) {
    v3_addscl(pt, cpos, axis_w, tstar);
    n += ora_sphere_box(c + n, pt, r, bpos, bmat, bsize, margin);
    /* plus the nearer end cap when it is also within the margin */
    v3_addscl(pt, cpos, axis_w, dpos <= dneg ? len : -len);
    if (n < 2) n += ora_sphere_box(c + n, pt, r, bpos, bmat, bsize, margin);
  } else {
    v3_addscl(pt, cpos, axis_w, len);
    n += ora_sphere_box(c + n, pt, r, bpos, bmat, bsize, margin);
    v3_addscl(pt, cpos, axis_w, -len);
    n += ora_sphere_box(c + n, pt, r, bpos, bmat, bsize, margin);
  }
  for (int k = 0; k < n; k++) v3_copy(c[k].frame + 3, axis_w);
  return n;
}

/* ---- ray casts (for the rangefinder): distance along the unit ray, or -1 */
static inline double ora_ray_plane(const double* gpos, const double* gmat, const double* gsize, const double* pnt,
                                   const double* vec) {
  double n[3] = {gmat[2], gmat[5], gmat[8]}, rel[3];
  v3_sub(rel, pnt, gpos);
  double denom = v3_dot(vec, n);
  if (denom > -ORA_MINVAL) return -1.0; /* only the front side is hit */
  double x = -v3_dot(rel, n) / denom;
  if (x < 0) return -1.0;
  /* finite planes clip to their half sizes */
  double hit[3], lx[3] = {gmat[0], gmat[3], gmat[6]}, ly[3] = {gmat[1], gmat[4], gmat[7]};
  v3_addscl(hit, rel, vec, x);
  if (gsize[0] > 0 && fabs(v3_dot(hit, lx)) > gsize[0]) return -1.0;
  if (gsize[1] > 0 && fabs(v3_dot(hit, ly)) > gsize[1]) return -1.0;
  return x;
}
static inline double ora_ray_sphere_at(const double* center, double r, const double* pnt, const double* vec) {
  double rel[3];
  v3_sub(rel, pnt, center);
  double b = v3_dot(vec, rel), cc = v3_dot(rel, rel) - r * r;
  double det = b * b - cc;
  if (det < 0) return -1.0;
  double sq = sqrt(det);
  double x0 = -b - sq, x1 = -b + sq;
  if (x0 >= 0) return x0;
  if (x1 >= 0) return x1;
  return -1.0;
}
static inline double ora_ray_capsule(const double* gpos, const double* gmat, const double* gsize, const double* pnt,
                                     const double* vec) {
  double axis[3] = {gmat[2], gmat[5], gmat[8]}, rel[3];
  double r = gsize[0], len = gsize[1], best = -1.0;
  v3_sub(rel, pnt, gpos);
  /* infinite cylinder about the axis, then clip to the segment */
  double va = v3_dot(vec, axis), ra = v3_dot(rel, axis);
  double vp[3], rp[3];
  v3_addscl(vp, vec, axis, -va);
  v3_addscl(rp, rel, axis, -ra);
  double a = v3_dot(vp, vp), b = v3_dot(vp, rp), cc = v3_dot(rp, rp) - r * r;
  if (a > ORA_MINVAL) {
    double det = b * b - a * cc;
    if (det >= 0) {
      double sq = sqrt(det);
      double xs[2] = {(-b - sq) / a, (-b + sq) / a};
      for (int k = 0; k < 2; k++) {
        double x = xs[k];
        if (x >= 0 && fabs(ra + x * va) <= len && (best < 0 || x < best)) best = x;
      }
    }
  }
  for (int s = -1; s <= 1; s += 2) {
    double cap[3];
    v3_addscl(cap, gpos, axis, s * len);
    double x = ora_ray_sphere_at(cap, r, pnt, vec);
    if (x >= 0) {
      /* keep only hits on the outer hemisphere */
      double h = ra + x * va;
      if (s * h >= len && (best < 0 || x < best)) best = x;
    }
  }
  return best;
}
static inline double ora_ray_box(const double* gpos, const double* gmat, const double* gsize, const double* pnt,
                                 const double* vec) {
  double rel[3], lp[3], lv[3];
  v3_sub(rel, pnt, gpos);
  m3_mulTv(lp, gmat, rel);
  m3_mulTv(lv, gmat, vec);
  double best = -1.0;
  for (int k = 0; k < 3; k++) {
    if (fabs(lv[k]) < ORA_MINVAL) continue;
    for (int s = -1; s <= 1; s += 2) {
      double x = (s * gsize[k] - lp[k]) / lv[k];
      if (x < 0) continue;
      int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
      double h1 = lp[k1] + x * lv[k1], h2 = lp[k2] + x * lv[k2];
      if (fabs(h1) <= gsize[k1] && fabs(h2) <= gsize[k2] && (best < 0 || x < best)) best = x;
    }
  }
  return best;
}
static inline double ora_ray_geom(int type, const double* gpos, const double* gmat, const double* gsize,
                                  const double* pnt, const double* vec) {
  switch (type) {
    case ORA_GEOM_PLANE: return ora_ray_plane(gpos, gmat, gsize, pnt, vec);
    case ORA_GEOM_SPHERE: return ora_ray_sphere_at(gpos, gsize[0], pnt, vec);
    case ORA_GEOM_CAPSULE: return ora_ray_capsule(gpos, gmat, gsize, pnt, vec);
    case ORA_GEOM_BOX: return ora_ray_box(gpos, gmat, gsize, pnt, vec);
    default: return -1.0;
  }
}

/* outward surface normal of a geom at a point on its surface (for the ray-cast camera's shading) */
static inline void ora_geom_normal(int type, const double* gpos, const double* gmat, const double* gsize,
                                   const double* hit, double* n) {
  double rel[3];
  v3_sub(rel, hit, gpos);
  if (type == ORA_GEOM_PLANE) { n[0] = gmat[2]; n[1] = gmat[5]; n[2] = gmat[8]; return; }
  if (type == ORA_GEOM_SPHERE) { v3_copy(n, rel); v3_normalize(n); return; }
  if (type == ORA_GEOM_CAPSULE) {
    double axis[3] = {gmat[2], gmat[5], gmat[8]};
    double h = v3_dot(rel, axis);
    if (h > gsize[1]) h = gsize[1];
    if (h < -gsize[1]) h = -gsize[1];
    v3_addscl(n, rel, axis, -h);
    v3_normalize(n);
    return;
  }
  /* box: the face whose normalised coordinate is largest */
  double loc[3], best = -1.0;
  int face = 0;
  m3_mulTv(loc, gmat, rel);
  for (int k = 0; k < 3; k++) {
    double a = fabs(loc[k]) / gsize[k];
    if (a > best) { best = a; face = k; }
  }
  double sg = loc[face] >= 0 ? 1.0 : -1.0;
  n[0] = sg * gmat[face]; n[1] = sg * gmat[3 + face]; n[2] = sg * gmat[6 + face];
}

#endif
