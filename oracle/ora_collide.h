/* ORACLE -- TEST INFRASTRUCTURE ONLY (see ora_math.h for the provenance statement).
 *
 * Narrow-phase contact generation for the primitive pairs the shipped levels contain
 * (plane, sphere, capsule, box).  Conventions follow MuJoCo's documented contact model:
 * the normal points from geom1 to geom2, dist<0 is penetration, pos is the midpoint between
 * the two surfaces, and the pair is ordered so that type(geom1) <= type(geom2).
 * The capsule-box routine is a from-scratch geometric construction (closest point of the
 * segment to the box, plus end caps), not a restatement of MuJoCo's feature walk.  The box-box routine
 * is likewise this repo's own: separating-axis test over the 15 axes, then either the clipped face
 * manifold (incident face against the reference face, up to 8 points) or one edge-edge contact.
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


/* ---- box against box.
 * Separating-axis test over the 6 face normals and the 9 edge cross products; the axis of least penetration decides.
 * A face axis: the manifold is the incident face (the face of the other box most anti-parallel to the reference normal)
 * clipped against the reference face's rectangle -- written as 12 independent candidates so that the device evaluates
 * them one per lane: for every incident edge its entry point into the rectangle (the edge's start vertex when that lies
 * inside) and, where the edge leaves the rectangle before its end, its exit point (items 0..7); the four corners of the
 * reference face that lie strictly inside the incident face's projection (items 8..11).  An edge axis (only when it
 * separates distinctly better than every face axis): one contact between the closest points of the two supporting edges
 * (item 12).  Candidates farther than the margin are dropped.  ora_box_box_item returns 1 when item k yields a contact. */
static inline int ora_box_box_item(ora_rawcon* c, int k, const double* p1, const double* m1, const double* s1,
                                   const double* p2, const double* m2, const double* s2, double margin) {
  double A[3][3], B[3][3], d[3], dA[3], dB[3], C[3][3], aC[3][3];
  for (int i = 0; i < 3; i++)
    for (int r = 0; r < 3; r++) { A[i][r] = m1[3 * r + i]; B[i][r] = m2[3 * r + i]; }
  v3_sub(d, p2, p1);
  for (int i = 0; i < 3; i++) { dA[i] = v3_dot(d, A[i]); dB[i] = v3_dot(d, B[i]); }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { C[i][j] = v3_dot(A[i], B[j]); aC[i][j] = fabs(C[i][j]); }
  double best = -1e300;
  int code = -1;
  for (int i = 0; i < 3; i++) {
    double rb = s2[0] * aC[i][0] + s2[1] * aC[i][1] + s2[2] * aC[i][2];
    double sep = fabs(dA[i]) - (s1[i] + rb);
    if (sep > best) { best = sep; code = i; }
  }
  for (int j = 0; j < 3; j++) {
    double ra = s1[0] * aC[0][j] + s1[1] * aC[1][j] + s1[2] * aC[2][j];
    double sep = fabs(dB[j]) - (ra + s2[j]);
    if (sep > best) { best = sep; code = 3 + j; }
  }
  if (best > margin) return 0;
  double ebest = -1e300, einv = 0;
  int ecode = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double L[3];
      v3_cross(L, A[i], B[j]);
      double l2 = v3_dot(L, L);
      if (l2 < 1e-10) continue;
      double inv = 1.0 / sqrt(l2);
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double ra = (s1[i1] * aC[i2][j] + s1[i2] * aC[i1][j]) * inv;
      double rb = (s2[j1] * aC[i][j2] + s2[j2] * aC[i][j1]) * inv;
      double sep = fabs(v3_dot(d, L) * inv) - (ra + rb);
      if (sep > ebest) { ebest = sep; ecode = 3 * i + j; einv = inv; }
    }
  if (ecode >= 0 && ebest > margin) return 0;
  int use_edge = ecode >= 0 && ebest > best + 0.05 * fabs(best) + 1e-9;
  v3_zero(c->frame + 3);
  if (use_edge) {
    if (k != 12) return 0;
    int i = ecode / 3, j = ecode % 3;
    double L[3], e1[3], e2[3], w[3];
    v3_cross(L, A[i], B[j]);
    v3_scl(L, L, einv);
    if (v3_dot(d, L) < 0) v3_scl(L, L, -1.0);
    v3_copy(e1, p1);
    v3_copy(e2, p2);
    for (int q = 0; q < 3; q++) {
      if (q != i) v3_addscl(e1, e1, A[q], (v3_dot(A[q], L) > 0 ? 1.0 : -1.0) * s1[q]);
      if (q != j) v3_addscl(e2, e2, B[q], (v3_dot(B[q], L) > 0 ? -1.0 : 1.0) * s2[q]);
    }
    v3_sub(w, e1, e2);
    double b = C[i][j], dd = v3_dot(A[i], w), ee = v3_dot(B[j], w), den = 1.0 - b * b;
    double ta = (b * ee - dd) / den, tb = (ee - b * dd) / den;
    if (ta > s1[i]) ta = s1[i]; if (ta < -s1[i]) ta = -s1[i];
    if (tb > s2[j]) tb = s2[j]; if (tb < -s2[j]) tb = -s2[j];
    double c1[3], c2[3], dif[3];
    v3_addscl(c1, e1, A[i], ta);
    v3_addscl(c2, e2, B[j], tb);
    v3_sub(dif, c2, c1);
    double dist = v3_dot(dif, L);
    if (dist > margin) return 0;
    c->dist = dist;
    v3_copy(c->frame, L);
    v3_add(c->pos, c1, c2);
    v3_scl(c->pos, c->pos, 0.5);
    return 1;
  }
  if (k >= 12) return 0;
  /* reference / incident box */
  const int ref1 = code < 3, ri = ref1 ? code : code - 3;
  const double (*RA)[3] = ref1 ? A : B, (*IA)[3] = ref1 ? B : A;
  const double *pR = ref1 ? p1 : p2, *pI = ref1 ? p2 : p1, *sR = ref1 ? s1 : s2, *sI = ref1 ? s2 : s1;
  double sgn = ref1 ? (dA[ri] >= 0 ? 1.0 : -1.0) : (dB[ri] >= 0 ? -1.0 : 1.0);
  double n[3];
  v3_scl(n, RA[ri], sgn);
  double cj[3] = {v3_dot(IA[0], n), v3_dot(IA[1], n), v3_dot(IA[2], n)};
  int ii = 0;
  if (fabs(cj[1]) > fabs(cj[ii])) ii = 1;
  if (fabs(cj[2]) > fabs(cj[ii])) ii = 2;
  double msgn = cj[ii] > 0 ? -1.0 : 1.0;
  int iu = (ii + 1) % 3, iv = (ii + 2) % 3, ru = (ri + 1) % 3, rv = (ri + 2) % 3;
  double cI[3], cR[3], eu[3], ev[3], V[4][3];
  v3_addscl(cI, pI, IA[ii], msgn * sI[ii]);
  v3_addscl(cR, pR, n, sR[ri]);
  v3_scl(eu, IA[iu], sI[iu]);
  v3_scl(ev, IA[iv], sI[iv]);
  for (int q = 0; q < 4; q++) {
    double su = (q == 0 || q == 3) ? 1.0 : -1.0, sv = q < 2 ? 1.0 : -1.0;
    v3_addscl(V[q], cI, eu, su);
    v3_addscl(V[q], V[q], ev, sv);
  }
  const double *tu = RA[ru], *tv = RA[rv];
  double hu = sR[ru], hv = sR[rv];
  double X[4], Y[4];
  for (int q = 0; q < 4; q++) {
    double rel[3];
    v3_sub(rel, V[q], cR);
    X[q] = v3_dot(rel, tu); Y[q] = v3_dot(rel, tv);
  }
  double nout = ref1 ? 1.0 : -1.0;
  if (k < 8) {
    int e = k >> 1, ex = k & 1, f = (e + 1) & 3;
    double xa = X[e], ya = Y[e], dx = X[f] - X[e], dy = Y[f] - Y[e];
    double t0 = 0.0, t1 = 1.0;
    int ok = 1;
    double pp[4] = {-dx, dx, -dy, dy}, qq[4] = {xa + hu, hu - xa, ya + hv, hv - ya};
    for (int b = 0; b < 4; b++) {
      if (pp[b] == 0.0) { if (qq[b] < 0.0) ok = 0; continue; }
      double r = qq[b] / pp[b];
      if (pp[b] < 0.0) { if (r > t0) t0 = r; } else { if (r < t1) t1 = r; }
    }
    if (!ok || t0 > t1) return 0;
    if (ex && !(t1 < 1.0)) return 0;
    double t = ex ? t1 : t0, P[3], dP[3], rel[3];
    v3_sub(dP, V[f], V[e]);
    v3_addscl(P, V[e], dP, t);
    v3_sub(rel, P, cR);
    double depth = v3_dot(rel, n);
    if (depth > margin) return 0;
    c->dist = depth;
    v3_scl(c->frame, n, nout);
    v3_addscl(c->pos, P, n, -0.5 * depth);
    return 1;
  }
  /* items 8..11: corners of the reference face strictly inside the incident face's projection */
  int q = k - 8;
  double x = ((q == 0 || q == 3) ? 1.0 : -1.0) * hu, y = (q < 2 ? 1.0 : -1.0) * hv;
  int pos = 0, neg = 0;
  for (int e = 0; e < 4; e++) {
    int f = (e + 1) & 3;
    double w = (X[f] - X[e]) * (y - Y[e]) - (Y[f] - Y[e]) * (x - X[e]);
    pos += w > 0.0; neg += w < 0.0;
  }
  if (pos != 4 && neg != 4) return 0;
  double mI[3], Vr[3], rel[3];
  v3_scl(mI, IA[ii], msgn);
  double nm = v3_dot(n, mI);
  if (nm > -1e-9) return 0;
  v3_addscl(Vr, cR, tu, x);
  v3_addscl(Vr, Vr, tv, y);
  v3_sub(rel, cI, Vr);
  double lam = v3_dot(rel, mI) / nm;
  if (lam > margin) return 0;
  c->dist = lam;
  v3_scl(c->frame, n, nout);
  v3_addscl(c->pos, Vr, n, 0.5 * lam);
  return 1;
}

static inline int ora_box_box(ora_rawcon* c, const double* p1, const double* m1, const double* s1, const double* p2,
                              const double* m2, const double* s2, double margin) {
  int n = 0;
  for (int k = 0; k < 13 && n < 8; k++) n += ora_box_box_item(c + n, k, p1, m1, s1, p2, m2, s2, margin);
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
