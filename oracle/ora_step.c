/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Single-environment, single-threaded fp64 restatement of what the reference gets from
 *   mj.MjModel.from_xml_path / mj.MjData          (mujoco_parent.py:126-127)
 *   mj.mj_resetData + mj.mj_forward               (mujoco_parent.py:349-350, 354-355)
 *   mj.mj_step                                    (mujoco_parent.py:335, 362)
 * for the MJCF subset of the shipped levels.  mujoco==2.3.3 (requirements.txt:45) is a
 * third-party dependency whose source is absent from /root/reference and which is not
 * installable here; the stage list follows MuJoCo's published pipeline:
 *   fwdPosition  : kinematics, comPos, crb, factorM, collision, makeConstraint
 *   fwdVelocity  : comVel, passive, rne
 *   fwdActuation : motor forces
 *   fwdAcceleration, fwdConstraint (PGS on the pyramidal-cone dual), sensors
 *   Euler        : semi-implicit, joint damping integrated implicitly
 * Solver note: the levels set no <option>, so real MuJoCo would default to the
 * Newton solver; BASELINE.json's north_star specifies PGS, and this oracle is PGS.
 *
 * PARITY UNPINNED for physics values (see ora_math.h).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ora_collide.h"
#include "ora_layout.h"

#define ORA_MINIMP 0.0001
#define ORA_MAXIMP 0.9999

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { SENS_TOUCH = 0, SENS_ACCELEROMETER = 1, SENS_RANGEFINDER = 2, SENS_FRAMEXAXIS = 3, SENS_FRAMEYAXIS = 4, SENS_FRAMEZAXIS = 5 };
enum { EFC_LIMIT = 0, EFC_CONTACT = 1 };

typedef struct ora_model {
#define X(name) int name;
  ORA_SIZE_FIELDS(X)
#undef X
#define X(name) double name;
  ORA_OPT_FIELDS(X)
#undef X
#define X(name, count) const double* name;
  ORA_F64_FIELDS(X)
#undef X
#define X(name, count) const int32_t* name;
  ORA_I32_FIELDS(X)
#undef X
  void* storage;
} ora_model;

typedef struct ora_contact {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5];
  int geom1, geom2, dim, efc_address;
} ora_contact;

typedef struct ora_data {
  double time;
  double *qpos, *qvel, *ctrl, *qacc_warmstart;
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis;
  double *geom_xpos, *geom_xmat, *site_xpos, *site_xmat, *cam_xpos, *cam_xmat;
  double *subtree_com, *cinert, *crb, *cdof, *cdof_dot, *cvel, *cacc, *cfrc;
  double *qM, *qLD, *qLDiagInv, *qMdense;
  double *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc;
  double *sensordata;
  int ncon, nefc, solver_niter, warn_con, warn_efc;
  ora_contact* contact;
  int *efc_type, *efc_id;
  double *efc_J, *efc_pos, *efc_margin, *efc_diagApprox, *efc_R, *efc_D, *efc_vel, *efc_aref, *efc_b,
      *efc_force, *efc_AR, *efc_KBIP;
  double* scratch;
} ora_data;

/* ------------------------------------------------------------------ model / data lifetime */
ora_model* ora_model_create(const void* blob, size_t nbytes) {
  const char* p = (const char*)blob;
  const int32_t* head = (const int32_t*)p;
  if (nbytes < 8 || head[0] != (int32_t)ORA_BLOB_MAGIC || head[1] != ORA_BLOB_VERSION) return NULL;
  ora_model* m = (ora_model*)calloc(1, sizeof(ora_model));
  m->storage = malloc(nbytes);
  memcpy(m->storage, blob, nbytes);
  p = (const char*)m->storage + 8;
  const int32_t* sz = (const int32_t*)p;
  int k = 0;
#define X(name) m->name = sz[k++];
  ORA_SIZE_FIELDS(X)
#undef X
  p += 4 * ORA_NSIZES;
  const double* op = (const double*)p;
  k = 0;
#define X(name) m->name = op[k++];
  ORA_OPT_FIELDS(X)
#undef X
  p += 8 * ORA_NOPTS;
  int nq = m->nq, nv = m->nv, nu = m->nu, nbody = m->nbody, njnt = m->njnt, ngeom = m->ngeom, nsite = m->nsite, nM = m->nM, ndesc = m->ndesc, nchild = m->nchild, ntree = m->ntree,
      ncam = m->ncam, nsensor = m->nsensor, npair = m->npair, nfactor = m->nfactor, ntab = m->ntab, nchunk = m->nchunk, ntp = m->ntp, nlight = m->nlight;
  (void)nfactor; (void)ntab; (void)nchunk; (void)ntp; (void)nlight;
  (void)nM; (void)ndesc; (void)nchild; (void)ntree; (void)nq; (void)nv; (void)nu; (void)nbody; (void)njnt; (void)ngeom; (void)nsite; (void)ncam; (void)nsensor; (void)npair;
#define X(name, count) m->name = (const double*)p; p += 8 * (size_t)(count);
  ORA_F64_FIELDS(X)
#undef X
#define X(name, count) m->name = (const int32_t*)p; p += 4 * (size_t)(((count) + 1) & ~1);
  ORA_I32_FIELDS(X)
#undef X
  if ((size_t)(p - (const char*)m->storage) != nbytes) {
    free(m->storage);
    free(m);
    return NULL;
  }
  return m;
}

void ora_model_destroy(ora_model* m) {
  if (!m) return;
  free(m->storage);
  free(m);
}

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

ora_data* ora_data_create(const ora_model* m) {
  ora_data* d = (ora_data*)calloc(1, sizeof(ora_data));
  int nb = m->nbody, nv = m->nv, nj = m->njnt, ng = m->ngeom, njmax = m->njmax;
  d->qpos = dalloc(m->nq); d->qvel = dalloc(nv); d->ctrl = dalloc(m->nu); d->qacc_warmstart = dalloc(nv);
  d->xpos = dalloc(nb * 3); d->xquat = dalloc(nb * 4); d->xmat = dalloc(nb * 9); d->xipos = dalloc(nb * 3);
  d->ximat = dalloc(nb * 9); d->xanchor = dalloc(nj * 3); d->xaxis = dalloc(nj * 3);
  d->geom_xpos = dalloc(ng * 3); d->geom_xmat = dalloc(ng * 9);
  d->site_xpos = dalloc(m->nsite * 3); d->site_xmat = dalloc(m->nsite * 9);
  d->cam_xpos = dalloc(m->ncam * 3); d->cam_xmat = dalloc(m->ncam * 9);
  d->subtree_com = dalloc(nb * 3); d->cinert = dalloc(nb * 10); d->crb = dalloc(nb * 10);
  d->cdof = dalloc(nv * 6); d->cdof_dot = dalloc(nv * 6); d->cvel = dalloc(nb * 6); d->cacc = dalloc(nb * 6);
  d->cfrc = dalloc(nb * 6);
  d->qM = dalloc(m->nM); d->qLD = dalloc(m->nM); d->qLDiagInv = dalloc(nv); d->qMdense = dalloc(nv * nv);
  d->qfrc_bias = dalloc(nv); d->qfrc_passive = dalloc(nv); d->qfrc_actuator = dalloc(nv);
  d->qfrc_smooth = dalloc(nv); d->qacc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv); d->qacc = dalloc(nv);
  d->sensordata = dalloc(m->nsensordata);
  d->contact = (ora_contact*)calloc(m->nconmax ? m->nconmax : 1, sizeof(ora_contact));
  d->efc_type = (int*)calloc(njmax ? njmax : 1, sizeof(int));
  d->efc_id = (int*)calloc(njmax ? njmax : 1, sizeof(int));
  d->efc_J = dalloc((size_t)njmax * nv); d->efc_pos = dalloc(njmax); d->efc_margin = dalloc(njmax);
  d->efc_diagApprox = dalloc(njmax); d->efc_R = dalloc(njmax); d->efc_D = dalloc(njmax); d->efc_vel = dalloc(njmax);
  d->efc_aref = dalloc(njmax); d->efc_b = dalloc(njmax); d->efc_force = dalloc(njmax);
  d->efc_AR = dalloc((size_t)njmax * njmax); d->efc_KBIP = dalloc(njmax * 4);
  d->scratch = dalloc((size_t)njmax * nv + 16 * (size_t)nv + 4 * (size_t)njmax + 64);
  return d;
}

void ora_data_destroy(ora_data* d) {
  if (!d) return;
  double** ptrs[] = {&d->qpos, &d->qvel, &d->ctrl, &d->qacc_warmstart, &d->xpos, &d->xquat, &d->xmat, &d->xipos,
                     &d->ximat, &d->xanchor, &d->xaxis, &d->geom_xpos, &d->geom_xmat, &d->site_xpos, &d->site_xmat,
                     &d->cam_xpos, &d->cam_xmat, &d->subtree_com, &d->cinert, &d->crb, &d->cdof, &d->cdof_dot,
                     &d->cvel, &d->cacc, &d->cfrc, &d->qM, &d->qLD, &d->qLDiagInv, &d->qMdense, &d->qfrc_bias,
                     &d->qfrc_passive, &d->qfrc_actuator, &d->qfrc_smooth, &d->qacc_smooth, &d->qfrc_constraint,
                     &d->qacc, &d->sensordata, &d->efc_J, &d->efc_pos, &d->efc_margin, &d->efc_diagApprox, &d->efc_R,
                     &d->efc_D, &d->efc_vel, &d->efc_aref, &d->efc_b, &d->efc_force, &d->efc_AR, &d->efc_KBIP,
                     &d->scratch};
  for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(*ptrs[i]);
  free(d->contact); free(d->efc_type); free(d->efc_id);
  free(d);
}

/* ------------------------------------------------------------------ position stage */
static void ora_kinematics(const ora_model* m, ora_data* d) {
  v3_zero(d->xpos); v3_zero(d->xipos);
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  q_to_mat(d->xmat, d->xquat);
  q_to_mat(d->ximat, d->xquat);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    double pos[3], quat[4], tmp[3];
    m3_mulv(tmp, d->xmat + 9 * p, m->body_pos + 3 * b);
    v3_add(pos, d->xpos + 3 * p, tmp);
    q_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k, qa = m->jnt_qposadr[j];
      if (m->jnt_type[j] == JNT_FREE) {
        v3_copy(pos, d->qpos + qa);
        memcpy(quat, d->qpos + qa + 3, 4 * sizeof(double));
        q_normalize(quat);
        v3_copy(d->xanchor + 3 * j, pos);
        q_rotv(d->xaxis + 3 * j, quat, m->jnt_axis + 3 * j);
      } else {
        double anchor[3], axis[3];
        q_rotv(tmp, quat, m->jnt_pos + 3 * j);
        v3_add(anchor, pos, tmp);
        q_rotv(axis, quat, m->jnt_axis + 3 * j);
        v3_copy(d->xanchor + 3 * j, anchor);
        v3_copy(d->xaxis + 3 * j, axis);
        double q = d->qpos[qa] - m->qpos0[qa];
        if (m->jnt_type[j] == JNT_HINGE) {
          double qloc[4], qnew[4];
          q_axis_angle(qloc, m->jnt_axis + 3 * j, q);
          q_mul(qnew, quat, qloc);
          memcpy(quat, qnew, sizeof(qnew));
          q_rotv(tmp, quat, m->jnt_pos + 3 * j);
          v3_sub(pos, anchor, tmp);
        } else {
          v3_addscl(pos, pos, axis, q);
        }
      }
    }
    q_normalize(quat);
    v3_copy(d->xpos + 3 * b, pos);
    memcpy(d->xquat + 4 * b, quat, 4 * sizeof(double));
    q_to_mat(d->xmat + 9 * b, quat);
    m3_mulv(tmp, d->xmat + 9 * b, m->body_ipos + 3 * b);
    v3_add(d->xipos + 3 * b, pos, tmp);
    double iq[4];
    q_mul(iq, quat, m->body_iquat + 4 * b);
    q_to_mat(d->ximat + 9 * b, iq);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double tmp[3], gq[4];
    m3_mulv(tmp, d->xmat + 9 * b, m->geom_pos + 3 * g);
    v3_add(d->geom_xpos + 3 * g, d->xpos + 3 * b, tmp);
    q_mul(gq, d->xquat + 4 * b, m->geom_quat + 4 * g);
    q_to_mat(d->geom_xmat + 9 * g, gq);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double tmp[3], gq[4];
    m3_mulv(tmp, d->xmat + 9 * b, m->site_pos + 3 * s);
    v3_add(d->site_xpos + 3 * s, d->xpos + 3 * b, tmp);
    q_mul(gq, d->xquat + 4 * b, m->site_quat + 4 * s);
    q_to_mat(d->site_xmat + 9 * s, gq);
  }
  for (int s = 0; s < m->ncam; s++) {
    int b = m->cam_bodyid[s];
    double tmp[3], gq[4];
    m3_mulv(tmp, d->xmat + 9 * b, m->cam_pos + 3 * s);
    v3_add(d->cam_xpos + 3 * s, d->xpos + 3 * b, tmp);
    q_mul(gq, d->xquat + 4 * b, m->cam_quat + 4 * s);
    q_to_mat(d->cam_xmat + 9 * s, gq);
  }
}

static void inert_com(double* res, const double* inert, const double* mat, const double* dif, double mass) {
  /* rotate the principal inertia into world orientation, shift to the reference point */
  double t[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) t[3 * r + c] = mat[3 * r + c] * inert[c];
  double full[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      full[3 * r + c] = t[3 * r] * mat[3 * c] + t[3 * r + 1] * mat[3 * c + 1] + t[3 * r + 2] * mat[3 * c + 2];
  res[0] = full[0] + mass * (dif[1] * dif[1] + dif[2] * dif[2]);
  res[1] = full[4] + mass * (dif[0] * dif[0] + dif[2] * dif[2]);
  res[2] = full[8] + mass * (dif[0] * dif[0] + dif[1] * dif[1]);
  res[3] = full[1] - mass * dif[0] * dif[1];
  res[4] = full[2] - mass * dif[0] * dif[2];
  res[5] = full[5] - mass * dif[1] * dif[2];
  res[6] = mass * dif[0]; res[7] = mass * dif[1]; res[8] = mass * dif[2];
  res[9] = mass;
}

static void ora_com_pos(const ora_model* m, ora_data* d) {
  int nb = m->nbody;
  for (int b = 0; b < nb; b++) v3_scl(d->subtree_com + 3 * b, d->xipos + 3 * b, m->body_mass[b]);
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    v3_add(d->subtree_com + 3 * p, d->subtree_com + 3 * p, d->subtree_com + 3 * b);
  }
  for (int b = 0; b < nb; b++) {
    if (m->body_subtreemass[b] < ORA_MINVAL) v3_copy(d->subtree_com + 3 * b, d->xipos + 3 * b);
    else v3_scl(d->subtree_com + 3 * b, d->subtree_com + 3 * b, 1.0 / m->body_subtreemass[b]);
  }
  memset(d->cinert, 0, 10 * sizeof(double));
  for (int b = 1; b < nb; b++) {
    double off[3];
    v3_sub(off, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
    inert_com(d->cinert + 10 * b, m->body_inertia + 3 * b, d->ximat + 9 * b, off, m->body_mass[b]);
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    double off[3];
    v3_sub(off, d->subtree_com + 3 * m->body_rootid[b], d->xanchor + 3 * j);
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        double* c = d->cdof + 6 * (da + k);
        memset(c, 0, 6 * sizeof(double));
        c[3 + k] = 1.0;
      }
      for (int k = 0; k < 3; k++) {
        double* c = d->cdof + 6 * (da + 3 + k);
        double axis[3] = {d->xmat[9 * b + k], d->xmat[9 * b + 3 + k], d->xmat[9 * b + 6 + k]};
        v3_copy(c, axis);
        v3_cross(c + 3, axis, off);
      }
    } else if (m->jnt_type[j] == JNT_HINGE) {
      double* c = d->cdof + 6 * da;
      v3_copy(c, d->xaxis + 3 * j);
      v3_cross(c + 3, d->xaxis + 3 * j, off);
    } else {
      double* c = d->cdof + 6 * da;
      v3_zero(c);
      v3_copy(c + 3, d->xaxis + 3 * j);
    }
  }
}

static void ora_crb(const ora_model* m, ora_data* d) {
  int nb = m->nbody, nv = m->nv;
  memcpy(d->crb, d->cinert, (size_t)nb * 10 * sizeof(double));
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0)
      for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k];
  }
  memset(d->qM, 0, (size_t)m->nM * sizeof(double));
  for (int i = 0; i < nv; i++) {
    double buf[6];
    sp_inert_mulv(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    int adr = m->dof_Madr[i];
    d->qM[adr] = m->dof_armature[i];
    for (int j = i; j >= 0; j = m->dof_parentid[j]) d->qM[adr++] += sp_dot(d->cdof + 6 * j, buf);
  }
}

/* sparse L^T D L factorisation of a matrix in the dof_Madr layout; in place.  (mj_factorI divides every entry of the
 * pivot row by the pivot; here the entries are multiplied by the pivot's reciprocal, the value that ends up in
 * qLDiagInv anyway -- the same algorithm with one division per pivot, and the form the device code uses.) */
static void factor_sparse(const ora_model* m, double* ld, double* diaginv) {
  for (int k = m->nv - 1; k >= 0; k--) {
    int kk = m->dof_Madr[k], ki = kk + 1;
    const double rk = 1.0 / ld[kk];
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i], ki++) {
      double a = ld[ki] * rk;
      int ij = m->dof_Madr[i], n = m->dof_depth[i] + 1;
      for (int t = 0; t < n; t++) ld[ij + t] -= a * ld[ki + t];
      ld[ki] = a;
    }
    diaginv[k] = rk;
  }
}

static void solve_sparse(const ora_model* m, const double* ld, const double* diaginv, double* x) {
  int nv = m->nv;
  for (int i = nv - 1; i >= 0; i--) {
    int adr = m->dof_Madr[i] + 1;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[j] -= ld[adr++] * x[i];
  }
  for (int i = 0; i < nv; i++) x[i] *= diaginv[i];
  for (int i = 0; i < nv; i++) {
    int adr = m->dof_Madr[i] + 1;
    for (int j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) x[i] -= ld[adr++] * x[j];
  }
}

static void ora_factor_m(const ora_model* m, ora_data* d) {
  memcpy(d->qLD, d->qM, (size_t)m->nM * sizeof(double));
  factor_sparse(m, d->qLD, d->qLDiagInv);
}

/* translational Jacobian of a world point moving with `body`: 3 x nv, row-major */
static void ora_jac_point(const ora_model* m, const ora_data* d, double* jacp, double* jacr, const double* point,
                          int body) {
  int nv = m->nv;
  if (jacp) memset(jacp, 0, 3 * (size_t)nv * sizeof(double));
  if (jacr) memset(jacr, 0, 3 * (size_t)nv * sizeof(double));
  double off[3];
  v3_sub(off, point, d->subtree_com + 3 * m->body_rootid[body]);
  for (int i = m->body_lastdof[body]; i >= 0; i = m->dof_parentid[i]) {
    const double* c = d->cdof + 6 * i;
    if (jacr) { jacr[i] = c[0]; jacr[nv + i] = c[1]; jacr[2 * nv + i] = c[2]; }
    if (jacp) {
      double t[3];
      v3_cross(t, c, off);
      jacp[i] = c[3] + t[0]; jacp[nv + i] = c[4] + t[1]; jacp[2 * nv + i] = c[5] + t[2];
    }
  }
}

/* ------------------------------------------------------------------ collision */
static int add_contact(const ora_model* m, ora_data* d, const ora_rawcon* rc, int g1, int g2, double margin,
                       double gap) {
  if (d->ncon >= m->nconmax) { d->warn_con = 1; return 0; }
  ora_contact* c = d->contact + d->ncon++;
  c->dist = rc->dist;
  v3_copy(c->pos, rc->pos);
  memcpy(c->frame, rc->frame, 9 * sizeof(double));
  ora_make_frame(c->frame);
  c->includemargin = margin - gap;
  c->geom1 = g1; c->geom2 = g2;
  c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
  const double *f1 = m->geom_friction + 3 * g1, *f2 = m->geom_friction + 3 * g2;
  double f[3];
  for (int k = 0; k < 3; k++) f[k] = f1[k] > f2[k] ? f1[k] : f2[k];
  c->friction[0] = c->friction[1] = f[0]; c->friction[2] = f[1]; c->friction[3] = c->friction[4] = f[2];
  double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
  if (s1 >= ORA_MINVAL && s2 >= ORA_MINVAL) mix = s1 / (s1 + s2);
  else if (s1 < ORA_MINVAL && s2 < ORA_MINVAL) mix = 0.5;
  else mix = s1 < ORA_MINVAL ? 0.0 : 1.0;
  for (int k = 0; k < 2; k++) c->solref[k] = mix * m->geom_solref[2 * g1 + k] + (1 - mix) * m->geom_solref[2 * g2 + k];
  for (int k = 0; k < 5; k++) c->solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
  c->efc_address = -1;
  return 1;
}

static void ora_collision(const ora_model* m, ora_data* d) {
  d->ncon = 0;
  d->warn_con = 0;
  /* the candidate pairs in the order of the model compiler's list (mjcf._pair_layout: segments by kind of broad-phase
   * test, one block per pair of kinematic trees, padded to chunks of 64 with empty entries) -- the order of the contacts */
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom[2 * p], g2 = m->pair_geom[2 * p + 1];
    if (g1 < 0) continue;                    /* padding entry */
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    double margin = m->geom_margin[g1] > m->geom_margin[g2] ? m->geom_margin[g1] : m->geom_margin[g2];
    double gap = m->geom_gap[g1] > m->geom_gap[g2] ? m->geom_gap[g1] : m->geom_gap[g2];
    const double *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2;
    const double *m1 = d->geom_xmat + 9 * g1, *m2 = d->geom_xmat + 9 * g2;
    const double *s1 = m->geom_size + 3 * g1, *s2 = m->geom_size + 3 * g2;
    /* bounding-sphere rejection (planes have no bound: test the signed distance of the sphere instead) */
    if (t1 == ORA_GEOM_PLANE) {
      double n[3] = {m1[2], m1[5], m1[8]}, dif[3];
      v3_sub(dif, p2, p1);
      if (v3_dot(dif, n) > m->geom_rbound[g2] + margin) continue;
    } else {
      double dif[3];
      v3_sub(dif, p2, p1);
      double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
      if (v3_dot(dif, dif) > bound * bound) continue;
    }
    ora_rawcon rc[8];
    int n = 0;
    if (t1 == ORA_GEOM_PLANE && t2 == ORA_GEOM_SPHERE) n = ora_plane_sphere_raw(rc, p1, m1, p2, s2[0], margin);
    else if (t1 == ORA_GEOM_PLANE && t2 == ORA_GEOM_CAPSULE) n = ora_plane_capsule(rc, p1, m1, p2, m2, s2, margin);
    else if (t1 == ORA_GEOM_PLANE && t2 == ORA_GEOM_BOX) n = ora_plane_box(rc, p1, m1, p2, m2, s2, margin);
    else if (t1 == ORA_GEOM_SPHERE && t2 == ORA_GEOM_SPHERE) n = ora_sphere_sphere_raw(rc, p1, s1[0], p2, s2[0], margin);
    else if (t1 == ORA_GEOM_SPHERE && t2 == ORA_GEOM_CAPSULE) n = ora_sphere_capsule(rc, p1, s1[0], p2, m2, s2, margin);
    else if (t1 == ORA_GEOM_SPHERE && t2 == ORA_GEOM_BOX) n = ora_sphere_box(rc, p1, s1[0], p2, m2, s2, margin);
    else if (t1 == ORA_GEOM_CAPSULE && t2 == ORA_GEOM_CAPSULE) n = ora_capsule_capsule(rc, p1, m1, s1, p2, m2, s2, margin);
    else if (t1 == ORA_GEOM_CAPSULE && t2 == ORA_GEOM_BOX) n = ora_capsule_box(rc, p1, m1, s1, p2, m2, s2, margin);
    else if (t1 == ORA_GEOM_BOX && t2 == ORA_GEOM_BOX) n = ora_box_box(rc, p1, m1, s1, p2, m2, s2, margin);
    for (int k = 0; k < n; k++) add_contact(m, d, rc + k, g1, g2, margin, gap);
  }
}

/* ------------------------------------------------------------------ constraints */
static double impedance(const double* solimp, double pos, double margin) {
  double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  if (dmin < ORA_MINIMP) dmin = ORA_MINIMP; if (dmin > ORA_MAXIMP) dmin = ORA_MAXIMP;
  if (dmax < ORA_MINIMP) dmax = ORA_MINIMP; if (dmax > ORA_MAXIMP) dmax = ORA_MAXIMP;
  if (width < ORA_MINVAL) width = ORA_MINVAL;
  if (mid < ORA_MINIMP) mid = ORA_MINIMP; if (mid > ORA_MAXIMP) mid = ORA_MAXIMP;
  if (power < 1) power = 1;
  if (dmin == dmax || width <= ORA_MINVAL) return 0.5 * (dmin + dmax);
  double x = (pos - margin) / width;
  if (x < 0) x = -x;
  if (x >= 1) return dmax;
  if (x == 0) return dmin;
  double y;
  if (power == 1) y = x;
  else if (power == 2) {                                       /* the default exponent: a plain square */
    if (x <= mid) { double t = x / mid; y = t * t * mid; }
    else { double t = (1 - x) / (1 - mid); y = 1 - t * t * (1 - mid); }
  }
  else if (x <= mid) y = pow(x / mid, power) * mid;           /* a*x^p with a = 1/mid^(p-1) */
  else y = 1 - pow((1 - x) / (1 - mid), power) * (1 - mid);    /* 1 - b*(1-x)^p with b = 1/(1-mid)^(p-1) */
  return dmin + y * (dmax - dmin);
}

static int add_row(const ora_model* m, ora_data* d, int type, int id, const double* jrow, double pos, double margin,
                   double diag_approx, const double* solref, const double* solimp) {
  if (d->nefc >= m->njmax) { d->warn_efc = 1; return 0; }
  int i = d->nefc++, nv = m->nv;
  d->efc_type[i] = type; d->efc_id[i] = id;
  memcpy(d->efc_J + (size_t)i * nv, jrow, (size_t)nv * sizeof(double));
  d->efc_pos[i] = pos; d->efc_margin[i] = margin; d->efc_diagApprox[i] = diag_approx;
  double vel = 0;
  for (int k = 0; k < nv; k++) vel += jrow[k] * d->qvel[k];
  d->efc_vel[i] = vel;
  double imp = impedance(solimp, pos, margin);
  double dmax = solimp[1];
  if (dmax < ORA_MINIMP) dmax = ORA_MINIMP; if (dmax > ORA_MAXIMP) dmax = ORA_MAXIMP;
  double timeconst = solref[0], dampratio = solref[1];
  if (timeconst < 2 * m->timestep) timeconst = 2 * m->timestep;
  double kk = dmax * dmax * timeconst * timeconst * dampratio * dampratio;
  double K = 1.0 / (kk > ORA_MINVAL ? kk : ORA_MINVAL);
  double bb = dmax * timeconst;
  double B = 2.0 / (bb > ORA_MINVAL ? bb : ORA_MINVAL);
  d->efc_KBIP[4 * i] = K; d->efc_KBIP[4 * i + 1] = B; d->efc_KBIP[4 * i + 2] = imp; d->efc_KBIP[4 * i + 3] = 0;
  double R = (1 - imp) / imp * diag_approx;
  if (R < ORA_MINVAL) R = ORA_MINVAL;
  d->efc_R[i] = R;
  d->efc_aref[i] = -B * vel - K * imp * (pos - margin);
  return 1;
}

static void ora_make_constraint(const ora_model* m, ora_data* d) {
  int nv = m->nv;
  d->nefc = 0;
  d->warn_efc = 0;
  double* jrow = d->scratch;               /* nv */
  double* jp1 = jrow + nv;                 /* 3*nv */
  double* jp2 = jp1 + 3 * nv;              /* 3*nv */
  double* jf = jp2 + 3 * nv;               /* 3*nv: difference Jacobian in the contact frame */
  /* joint limits */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || (m->jnt_type[j] != JNT_HINGE && m->jnt_type[j] != JNT_SLIDE)) continue;
    double value = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - value);
      if (dist < margin) {
        memset(jrow, 0, (size_t)nv * sizeof(double));
        jrow[m->jnt_dofadr[j]] = -side;
        add_row(m, d, EFC_LIMIT, j, jrow, dist, margin, m->dof_invweight0[m->jnt_dofadr[j]], m->jnt_solref + 2 * j,
                m->jnt_solimp + 5 * j);
      }
    }
  }
  /* contacts: pyramidal cone, 2*(dim-1) rows each */
  for (int ci = 0; ci < d->ncon; ci++) {
    ora_contact* c = d->contact + ci;
    c->efc_address = -1;
    if (c->dist >= c->includemargin) continue;
    int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
    ora_jac_point(m, d, jp1, NULL, c->pos, b1);
    ora_jac_point(m, d, jp2, NULL, c->pos, b2);
    for (int r = 0; r < 3; r++)
      for (int k = 0; k < nv; k++) {
        double s = 0;
        for (int a = 0; a < 3; a++) s += c->frame[3 * r + a] * (jp2[a * nv + k] - jp1[a * nv + k]);
        jf[r * nv + k] = s;
      }
    double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
    int first = d->nefc;
    if (c->dim == 1) {
      if (add_row(m, d, EFC_CONTACT, ci, jf, c->dist, c->includemargin, tran, c->solref, c->solimp)) c->efc_address = first;
      continue;
    }
    int ok = 1;
    for (int k = 1; k < c->dim && k < 3 && ok; k++) {
      double mu = c->friction[k - 1];
      for (int sgn = 1; sgn >= -1 && ok; sgn -= 2) {
        for (int q = 0; q < nv; q++) jrow[q] = jf[q] + sgn * mu * jf[k * nv + q];
        ok = add_row(m, d, EFC_CONTACT, ci, jrow, c->dist, c->includemargin, tran + mu * mu * tran, c->solref,
                     c->solimp);
      }
    }
    if (!ok) { d->nefc = first; continue; }   /* a partially added pyramid is dropped as a whole */
    c->efc_address = first;
    /* pyramid regularisation: every edge gets 2*mu^2*R(first edge) */
    double Rpy = 2 * c->friction[0] * c->friction[0] * d->efc_R[first];
    if (Rpy < ORA_MINVAL) Rpy = ORA_MINVAL;
    for (int r = first; r < d->nefc; r++) d->efc_R[r] = Rpy;
  }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1.0 / d->efc_R[i];
}

/* ------------------------------------------------------------------ velocity stage */
static void ora_com_vel(const ora_model* m, ora_data* d) {
  memset(d->cvel, 0, 6 * sizeof(double));
  for (int b = 1; b < m->nbody; b++) {
    double cvel[6];
    memcpy(cvel, d->cvel + 6 * m->body_parentid[b], sizeof(cvel));
    int da = m->body_dofadr[b];
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k;
      if (m->jnt_type[j] == JNT_FREE) {
        for (int t = 0; t < 3; t++) {
          memset(d->cdof_dot + 6 * (da + t), 0, 6 * sizeof(double));
          for (int r = 0; r < 6; r++) cvel[r] += d->cdof[6 * (da + t) + r] * d->qvel[da + t];
        }
        da += 3;
        for (int t = 0; t < 3; t++) sp_cross_motion(d->cdof_dot + 6 * (da + t), cvel, d->cdof + 6 * (da + t));
        for (int t = 0; t < 3; t++)
          for (int r = 0; r < 6; r++) cvel[r] += d->cdof[6 * (da + t) + r] * d->qvel[da + t];
        da += 3;
      } else {
        sp_cross_motion(d->cdof_dot + 6 * da, cvel, d->cdof + 6 * da);
        for (int r = 0; r < 6; r++) cvel[r] += d->cdof[6 * da + r] * d->qvel[da];
        da++;
      }
    }
    memcpy(d->cvel + 6 * b, cvel, sizeof(cvel));
  }
}

/* recursive Newton-Euler; with_acc adds cdof*qacc (used after the solve, for the accelerometer) */
static void ora_rne(const ora_model* m, ora_data* d, int with_acc, double* result) {
  int nb = m->nbody, nv = m->nv;
  double* cacc = d->cacc;
  double* cfrc = d->cfrc;
  cacc[0] = cacc[1] = cacc[2] = 0;
  cacc[3] = -m->gravity_x; cacc[4] = -m->gravity_y; cacc[5] = -m->gravity_z;
  memset(cfrc, 0, 6 * sizeof(double));
  for (int b = 1; b < nb; b++) {
    int da = m->body_dofadr[b];
    double* a = cacc + 6 * b;
    memcpy(a, cacc + 6 * m->body_parentid[b], 6 * sizeof(double));
    for (int t = 0; t < m->body_dofnum[b]; t++)
      for (int r = 0; r < 6; r++) {
        a[r] += d->cdof_dot[6 * (da + t) + r] * d->qvel[da + t];
        if (with_acc) a[r] += d->cdof[6 * (da + t) + r] * d->qacc[da + t];
      }
    double t1[6], t2[6], t3[6];
    sp_inert_mulv(t1, d->cinert + 10 * b, a);
    sp_inert_mulv(t2, d->cinert + 10 * b, d->cvel + 6 * b);
    sp_cross_force(t3, d->cvel + 6 * b, t2);
    for (int r = 0; r < 6; r++) cfrc[6 * b + r] = t1[r] + t3[r];
  }
  if (!result) return;
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    if (p > 0)
      for (int r = 0; r < 6; r++) cfrc[6 * p + r] += cfrc[6 * b + r];
  }
  for (int i = 0; i < nv; i++) result[i] = sp_dot(d->cdof + 6 * i, cfrc + 6 * m->dof_bodyid[i]);
}

/* ------------------------------------------------------------------ solver */
static void mul_M_dense(const ora_model* m, const ora_data* d, double* dense) {
  int nv = m->nv;
  memset(dense, 0, (size_t)nv * nv * sizeof(double));
  for (int i = 0; i < nv; i++) {
    int adr = m->dof_Madr[i];
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      dense[i * nv + j] = d->qM[adr];
      dense[j * nv + i] = d->qM[adr];
      adr++;
    }
  }
}

static void ora_fwd_constraint(const ora_model* m, ora_data* d) {
  int nv = m->nv, nefc = d->nefc;
  d->solver_niter = 0;
  if (nefc == 0) {
    memcpy(d->qacc, d->qacc_smooth, (size_t)nv * sizeof(double));
    memcpy(d->qacc_warmstart, d->qacc_smooth, (size_t)nv * sizeof(double));
    memset(d->qfrc_constraint, 0, (size_t)nv * sizeof(double));
    return;
  }
  double* MinvJT = d->scratch;                    /* nefc x nv : row i = M^-1 J_i^T */
  double* AR = d->efc_AR;
  for (int i = 0; i < nefc; i++) {
    memcpy(MinvJT + (size_t)i * nv, d->efc_J + (size_t)i * nv, (size_t)nv * sizeof(double));
    solve_sparse(m, d->qLD, d->qLDiagInv, MinvJT + (size_t)i * nv);
  }
  for (int i = 0; i < nefc; i++)
    for (int j = 0; j < nefc; j++) {
      double s = 0;
      for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * MinvJT[(size_t)j * nv + k];
      AR[(size_t)i * nefc + j] = s + (i == j ? d->efc_R[i] : 0.0);
    }
  for (int i = 0; i < nefc; i++) {
    double s = 0;
    for (int k = 0; k < nv; k++) s += d->efc_J[(size_t)i * nv + k] * d->qacc_smooth[k];
    d->efc_b[i] = s - d->efc_aref[i];
  }
  /* warm start: forces implied by last step's acceleration, kept only if they beat zero */
  double* f = d->efc_force;
  for (int i = 0; i < nefc; i++) {
    double jar = -d->efc_aref[i];
    for (int k = 0; k < nv; k++) jar += d->efc_J[(size_t)i * nv + k] * d->qacc_warmstart[k];
    f[i] = jar < 0 ? -d->efc_D[i] * jar : 0.0;
  }
  double cost = 0;
  for (int i = 0; i < nefc; i++) {
    double s = 0;
    for (int j = 0; j < nefc; j++) s += AR[(size_t)i * nefc + j] * f[j];
    cost += 0.5 * f[i] * s + f[i] * d->efc_b[i];
  }
  if (cost > 0) memset(f, 0, (size_t)nefc * sizeof(double));
  /* projected Gauss-Seidel on  min 1/2 f'AR f + f'b,  f >= 0 */
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  while (iter < m->iterations) {
    double improvement = 0;
    for (int i = 0; i < nefc; i++) {
      double res = d->efc_b[i];
      for (int j = 0; j < nefc; j++) res += AR[(size_t)i * nefc + j] * f[j];
      double old = f[i], aii = AR[(size_t)i * nefc + i];
      double fn = old - res * (1.0 / aii);   /* the solver keeps the inverse diagonal and multiplies */
      if (fn < 0) fn = 0;
      double delta = fn - old;
      double change = 0.5 * delta * delta * aii + delta * res;
      if (change > 1e-10) { fn = old; change = 0; }
      f[i] = fn;
      improvement -= change;
    }
    iter++;
    if (improvement * scale < m->tolerance) break;
  }
  d->solver_niter = iter;
  for (int k = 0; k < nv; k++) {
    double s = 0;
    for (int i = 0; i < nefc; i++) s += d->efc_J[(size_t)i * nv + k] * f[i];
    d->qfrc_constraint[k] = s;
  }
  memcpy(d->qacc, d->qfrc_constraint, (size_t)nv * sizeof(double));
  solve_sparse(m, d->qLD, d->qLDiagInv, d->qacc);
  for (int k = 0; k < nv; k++) d->qacc[k] += d->qacc_smooth[k];
  memcpy(d->qacc_warmstart, d->qacc, (size_t)nv * sizeof(double));
}

/* ------------------------------------------------------------------ sensors */
static double contact_normal_force(const ora_data* d, const ora_contact* c) {
  if (c->efc_address < 0) return 0;
  if (c->dim == 1) return d->efc_force[c->efc_address];
  double s = 0;
  int rows = 2 * ((c->dim < 3 ? c->dim : 3) - 1);
  for (int r = 0; r < rows; r++) s += d->efc_force[c->efc_address + r];
  return s;
}

static void ora_sensors(const ora_model* m, ora_data* d) {
  int need_acc = 0;
  for (int s = 0; s < m->nsensor; s++) need_acc |= (m->sensor_type[s] == SENS_ACCELEROMETER);
  if (need_acc) ora_rne(m, d, 1, NULL);
  for (int s = 0; s < m->nsensor; s++) {
    int site = m->sensor_objid[s], adr = m->sensor_adr[s], body = m->site_bodyid[site];
    const double *sp = d->site_xpos + 3 * site, *sm = d->site_xmat + 9 * site;
    double cutoff = m->sensor_cutoff[s];
    double* out = d->sensordata + adr;
    switch (m->sensor_type[s]) {
      case SENS_RANGEFINDER: {
        double vec[3] = {sm[2], sm[5], sm[8]}, best = -1;
        for (int g = 0; g < m->ngeom; g++) {
          if (m->geom_bodyid[g] == body) continue;
          if (m->geom_rgba[4 * g + 3] == 0) continue;
          double x = ora_ray_geom(m->geom_type[g], d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, m->geom_size + 3 * g,
                                  sp, vec);
          if (x >= 0 && (best < 0 || x < best)) best = x;
        }
        out[0] = best;
        if (cutoff > 0 && out[0] > cutoff) out[0] = cutoff;
        break;
      }
      case SENS_TOUCH: {
        double total = 0;
        for (int ci = 0; ci < d->ncon; ci++) {
          const ora_contact* c = d->contact + ci;
          int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
          if (c->efc_address < 0 || (b1 != body && b2 != body)) continue;
          double fn = contact_normal_force(d, c);
          if (fn <= 0) continue;
          double ray[3];
          v3_scl(ray, c->frame, b2 == body ? -1.0 : 1.0);
          /* the force counts when the line through the contact point along the normal meets the site volume */
          if (ora_ray_sphere_at(sp, m->site_size[3 * site], c->pos, ray) >= 0) total += fn;
        }
        out[0] = total;
        if (cutoff > 0 && out[0] > cutoff) out[0] = cutoff;
        break;
      }
      case SENS_ACCELEROMETER: {
        /* spatial acceleration / velocity of the body at the site, in the site frame, plus the w x v term */
        const double* com = d->subtree_com + 3 * m->body_rootid[body];
        double off[3], acc[3], vel[3], t[3];
        v3_sub(off, sp, com);
        v3_cross(t, d->cacc + 6 * body, off);
        v3_add(acc, d->cacc + 6 * body + 3, t);
        v3_cross(t, d->cvel + 6 * body, off);
        v3_add(vel, d->cvel + 6 * body + 3, t);
        double wl[3], vl[3], al[3];
        m3_mulTv(wl, sm, d->cvel + 6 * body);
        m3_mulTv(vl, sm, vel);
        m3_mulTv(al, sm, acc);
        v3_cross(t, wl, vl);
        for (int k = 0; k < 3; k++) {
          out[k] = al[k] + t[k];
          if (cutoff > 0) { if (out[k] > cutoff) out[k] = cutoff; if (out[k] < -cutoff) out[k] = -cutoff; }
        }
        break;
      }
      case SENS_FRAMEXAXIS: case SENS_FRAMEYAXIS: case SENS_FRAMEZAXIS: {
        int col = m->sensor_type[s] - SENS_FRAMEXAXIS;
        for (int k = 0; k < 3; k++) out[k] = sm[3 * k + col];
        break;
      }
      default: break;
    }
  }
}

/* ------------------------------------------------------------------ pipeline */
void ora_forward(const ora_model* m, ora_data* d) {
  int nv = m->nv;
  ora_kinematics(m, d);
  ora_com_pos(m, d);
  ora_crb(m, d);
  ora_factor_m(m, d);
  ora_collision(m, d);
  ora_make_constraint(m, d);
  ora_com_vel(m, d);
  /* passive forces: joint damping, and the joint springs -stiffness (q - springref) of hinges and slides (mj_passive) */
  for (int i = 0; i < nv; i++) {
    double frc = -m->dof_damping[i] * d->qvel[i];
    frc -= m->dof_stiffness[i] * (d->qpos[m->dof_qposadr[i]] - m->dof_springref[i]);
    d->qfrc_passive[i] = frc;
  }
  ora_rne(m, d, 0, d->qfrc_bias);
  memset(d->qfrc_actuator, 0, (size_t)nv * sizeof(double));
  for (int u = 0; u < m->nu; u++) {
    double c = d->ctrl[u];
    if (m->act_ctrllimited[u]) {
      if (c < m->act_ctrlrange[2 * u]) c = m->act_ctrlrange[2 * u];
      if (c > m->act_ctrlrange[2 * u + 1]) c = m->act_ctrlrange[2 * u + 1];
    }
    d->qfrc_actuator[m->act_dofid[u]] += m->act_gear[u] * c;
  }
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  memcpy(d->qacc_smooth, d->qfrc_smooth, (size_t)nv * sizeof(double));
  solve_sparse(m, d->qLD, d->qLDiagInv, d->qacc_smooth);
  ora_fwd_constraint(m, d);
  ora_sensors(m, d);
  mul_M_dense(m, d, d->qMdense);
}

static void ora_integrate_pos(const ora_model* m, double* qpos, const double* qvel, double h) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * qvel[da + k];
      double w[3] = {qvel[da + 3], qvel[da + 4], qvel[da + 5]}, qrot[4], qn[4];
      double angle = h * v3_normalize(w);
      q_axis_angle(qrot, w, angle);
      q_normalize(qpos + qa + 3);
      q_mul(qn, qpos + qa + 3, qrot);
      memcpy(qpos + qa + 3, qn, sizeof(qn));
    } else {
      qpos[qa] += h * qvel[da];
    }
  }
}

/* 4th-order Runge-Kutta (<option integrator="RK4">, benchmarking/levels/Ant.xml:3), the classical tableau: the state
 * (qpos, qvel) is advanced with the weighted derivatives of four forward passes, the three later ones at trial states
 * X0 + h c F_prev (positions on the configuration manifold: quaternions by the exponential map).  Sensors belong to the
 * first pass (the step's own mj_forward); joint damping is explicit here (no implicit-damping solve).  Every pass leaves
 * its acceleration as the next pass's warm start, as ora_forward always does. */
static void ora_step_rk4(const ora_model* m, ora_data* d) {
  const int nv = m->nv, nq = m->nq;
  const double h = m->timestep;
  static const double Bw[4] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0}, Cn[3] = {0.5, 0.5, 1.0};
  double* keep = (double*)malloc(sizeof(double) * (size_t)(nq + 4 * nv + m->nsensordata + 1));
  double *x0q = keep, *x0v = x0q + nq, *accv = x0v + nv, *acca = accv + nv, *vtmp = acca + nv, *sens = vtmp + nv;
  for (int stage = 0; stage < 4; stage++) {
    ora_forward(m, d);
    if (stage == 0) {
      memcpy(x0q, d->qpos, sizeof(double) * (size_t)nq);
      memcpy(x0v, d->qvel, sizeof(double) * (size_t)nv);
      memcpy(sens, d->sensordata, sizeof(double) * (size_t)m->nsensordata);
      for (int i = 0; i < nv; i++) { accv[i] = Bw[0] * d->qvel[i]; acca[i] = Bw[0] * d->qacc[i]; }
    } else {
      for (int i = 0; i < nv; i++) { accv[i] = accv[i] + Bw[stage] * d->qvel[i]; acca[i] = acca[i] + Bw[stage] * d->qacc[i]; }
    }
    if (stage < 3) {
      const double c = Cn[stage];
      for (int i = 0; i < nv; i++) { vtmp[i] = c * d->qvel[i]; d->qvel[i] = x0v[i] + h * (c * d->qacc[i]); }
    } else {
      for (int i = 0; i < nv; i++) { vtmp[i] = accv[i]; d->qvel[i] = x0v[i] + h * acca[i]; }
    }
    memcpy(d->qpos, x0q, sizeof(double) * (size_t)nq);
    ora_integrate_pos(m, d->qpos, vtmp, h);
  }
  memcpy(d->sensordata, sens, sizeof(double) * (size_t)m->nsensordata);
  d->time += h;
  free(keep);
}

void ora_step(const ora_model* m, ora_data* d) {
  if (m->integrator == 1) { ora_step_rk4(m, d); return; }
  int nv = m->nv;
  double h = m->timestep;
  ora_forward(m, d);
  /* semi-implicit Euler; joint damping enters implicitly through (M + h*diag(damping)) */
  int damped = 0;
  for (int i = 0; i < nv; i++) damped |= (m->dof_damping[i] > 0);
  double* qacc = d->scratch;
  if (damped) {
    double* ld = d->scratch + nv;
    double* dinv = ld + m->nM;
    memcpy(ld, d->qM, (size_t)m->nM * sizeof(double));
    for (int i = 0; i < nv; i++) ld[m->dof_Madr[i]] += h * m->dof_damping[i];
    factor_sparse(m, ld, dinv);
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    solve_sparse(m, ld, dinv, qacc);
  } else {
    memcpy(qacc, d->qacc, (size_t)nv * sizeof(double));
  }
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  ora_integrate_pos(m, d->qpos, d->qvel, h);
  d->time += h;
}

void ora_reset(const ora_model* m, ora_data* d) {
  memcpy(d->qpos, m->qpos0, (size_t)m->nq * sizeof(double));
  memset(d->qvel, 0, (size_t)m->nv * sizeof(double));
  memset(d->ctrl, 0, (size_t)m->nu * sizeof(double));
  memset(d->qacc_warmstart, 0, (size_t)m->nv * sizeof(double));
  d->time = 0;
  ora_forward(m, d);
}

/* ------------------------------------------------------------------ camera (config 5)
 * Fixed body-mounted camera as a ray caster: the reference renders it with OpenGL (mujoco_parent.py:518-538,
 * mjv_updateScene + mjr_render + mjr_readPixels); no renderer is available here, so pixel parity with the reference
 * is UNPINNED.  Conventions kept from the reference: the camera looks along -z of its frame with +x right and +y up,
 * vertical field of view fovy, uint8 RGB, rows stored bottom-up as glReadPixels returns them, shape (W, H, 3).
 *
 * Shading (round 3): the OpenGL fixed-function lighting equation that MuJoCo's renderer drives, with the parameters
 * MuJoCo documents (XML reference: visual/headlight -- ambient 0.1, diffuse 0.4, specular 0.5, active; body/light --
 * directional false, pos 0 0 0, dir 0 0 -1, attenuation 1 0 0, cutoff 45, exponent 10, ambient 0, diffuse 0.7,
 * specular 0.3; asset/material -- specular 0.5, shininess 0.5, emission 0, which are also what a geom without a
 * material is drawn with).  Per colour channel, with the geom's rgba as ambient and diffuse material colour:
 *   c = emission * rgba
 *     + sum over lights  att * spot * ( amb_l * rgba + max(n.L, 0) * diff_l * rgba
 *                                       + [n.L > 0] * max(n.H, 0)^(128 * shininess) * spec_l * specular )
 * clamped to [0, 1] once, at the end (rgba itself is NOT clamped: the levels' "255 255 255" floor saturates as it does
 * in OpenGL).  Lights: the headlight -- directional, shining along the camera's viewing direction (L = the camera's
 * +z axis), att = spot = 1 -- and every <light> of the level: L = -dir for a directional one; else L points from the
 * surface point to the light, att = 1 / (k0 + k1 d + k2 d^2), and for cutoff < 180 the OpenGL spot factor
 * (-L . dir)^exponent inside the cone, 0 outside.  H = normalize(L + V) with V = the camera's +z axis (OpenGL's default
 * viewer at infinity), the scene's global ambient is 0.
 * Shadows (round 4): MuJoCo draws the shadow of every light with castshadow (body/light, default true; the headlight
 * casts none).  mjr_render does it with a depth map rendered from the light; the exact form of the same statement is a
 * shadow ray: from the surface point towards the light (for a positional light up to the light's position), and if any
 * other opaque geom lies on it the light's diffuse and specular terms are dropped for that point -- its ambient term
 * stays.  The geom the point lies on is not tested against its own point: all primitives here are convex, so it can only
 * hide the light where n.L <= 0, and there the two terms are zero anyway.  (What a depth map adds -- soft, resolution-
 * dependent edges -- is not reproduced.)
 * What is NOT reproduced: textures and
 * reflectance (the checker floor of Ant.xml), the skybox, anti-aliasing, fog / haze, transparency blending; OpenGL
 * evaluates the equation per vertex of the tessellated geoms and interpolates, this evaluates it per pixel; the viewer
 * model (local / at infinity) and the zero global ambient are read off OpenGL's defaults, not off MuJoCo's source. */
static void ora_shade(const ora_model* m, const ora_data* d, const double* cm, int hit, const double* p, const double* n,
                      unsigned char* px) {
  const double* rgba = m->geom_rgba + 4 * hit;
  const double spec_m = m->geom_matprop[3 * hit], shin = 128.0 * m->geom_matprop[3 * hit + 1], emis = m->geom_matprop[3 * hit + 2];
  const double V[3] = {cm[2], cm[5], cm[8]};        /* the camera's +z axis: towards the viewer */
  double col[3] = {emis * rgba[0], emis * rgba[1], emis * rgba[2]};
  for (int li = -1; li < m->nlight; li++) {
    double L[3], scale = 1.0, light_dist = 0.0;
    const double *amb, *dif, *spc;
    if (li < 0) {
      if (m->headlight[0] == 0) continue;
      L[0] = V[0]; L[1] = V[1]; L[2] = V[2];
      amb = m->headlight + 1; dif = m->headlight + 4; spc = m->headlight + 7;
    } else {
      const int b = m->light_bodyid[li];
      double pos[3], dir[3];
      m3_mulv(pos, d->xmat + 9 * b, m->light_pos + 3 * li);
      v3_add(pos, pos, d->xpos + 3 * b);
      m3_mulv(dir, d->xmat + 9 * b, m->light_dir + 3 * li);
      amb = m->light_ambient + 3 * li; dif = m->light_diffuse + 3 * li; spc = m->light_specular + 3 * li;
      if (m->light_directional[li]) {
        L[0] = -dir[0]; L[1] = -dir[1]; L[2] = -dir[2];
      } else {
        v3_sub(L, pos, p);
        double dist = sqrt(v3_dot(L, L));
        if (dist < ORA_MINVAL) continue;
        light_dist = dist;
        L[0] /= dist; L[1] /= dist; L[2] /= dist;
        const double* k = m->light_attenuation + 3 * li;
        scale = 1.0 / (k[0] + k[1] * dist + k[2] * dist * dist);
        if (m->light_cutoff[li] < 180.0) {
          double c = -v3_dot(L, dir);
          scale = c < cos(m->light_cutoff[li] * ORA_PI / 180.0) ? 0.0 : scale * pow(c > 0 ? c : 0, m->light_exponent[li]);
        }
      }
    }
    double nl = v3_dot(n, L), sp = 0;
    if (nl > 0 && scale > 0 && li >= 0 && m->light_castshadow[li]) {
      /* the shadow ray: is another opaque geom between the point and the light? */
      for (int g = 0; g < m->ngeom && nl > 0; g++) {
        if (g == hit || m->geom_rgba[4 * g + 3] == 0) continue;
        double x = ora_ray_geom(m->geom_type[g], d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, m->geom_size + 3 * g, p, L);
        if (x >= 0 && (m->light_directional[li] || x < light_dist)) nl = 0;
      }
    }
    if (nl > 0) {
      double H[3] = {L[0] + V[0], L[1] + V[1], L[2] + V[2]};
      double hn = sqrt(v3_dot(H, H));
      double nh = hn > ORA_MINVAL ? v3_dot(n, H) / hn : 0;
      sp = nh > 0 ? pow(nh, shin) : 0;
    } else {
      nl = 0;
    }
    for (int k = 0; k < 3; k++) col[k] += scale * (amb[k] * rgba[k] + nl * dif[k] * rgba[k] + sp * spc[k] * spec_m);
  }
  for (int k = 0; k < 3; k++) {
    double c = col[k] > 1 ? 1 : (col[k] < 0 ? 0 : col[k]);
    px[k] = (unsigned char)(255.0 * c + 0.5);
  }
}

/* Which frames: mjv_updateScene(model, data, ...) (mujoco_parent.py:533) reads the geom, camera and light frames out of
 * MjData as the last forward pass left them -- after mj_step (forward, then integrate) they are one integration older
 * than qpos, and the reference calls no mj_forward in between (mujoco_parent.py:333-336 -> 540-555).  So this draws the
 * frames d holds and runs no kinematics of its own; a caller that wrote qpos by hand calls ora_forward first. */
void ora_render(const ora_model* m, ora_data* d, int cam, int width, int height, unsigned char* out) {
  const double* cp = d->cam_xpos + 3 * cam;
  const double* cm = d->cam_xmat + 9 * cam;
  double t = tan(0.5 * m->cam_fovy[cam] * ORA_PI / 180.0), aspect = (double)width / (double)height;
  for (int r = 0; r < height; r++)
    for (int c = 0; c < width; c++) {
      double lx = (2.0 * (c + 0.5) / width - 1.0) * t * aspect, ly = (2.0 * (r + 0.5) / height - 1.0) * t;
      double loc[3] = {lx, ly, -1.0}, vec[3];
      m3_mulv(vec, cm, loc);
      v3_normalize(vec);
      double best = -1.0;
      int hit = -1;
      for (int g = 0; g < m->ngeom; g++) {
        if (m->geom_rgba[4 * g + 3] == 0) continue;
        double x = ora_ray_geom(m->geom_type[g], d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, m->geom_size + 3 * g, cp, vec);
        if (x >= 0 && (best < 0 || x < best)) { best = x; hit = g; }
      }
      unsigned char* px = out + 3 * ((size_t)r * width + c);
      if (hit < 0) { px[0] = px[1] = px[2] = 0; continue; }
      double p[3], n[3];
      v3_addscl(p, cp, vec, best);
      ora_geom_normal(m->geom_type[hit], d->geom_xpos + 3 * hit, d->geom_xmat + 9 * hit, m->geom_size + 3 * hit, p, n);
      ora_shade(m, d, cm, hit, p, n, px);
    }
}

/* The same rays, for the test harness: per pixel the geom hit (-1: none) and the distance along the ray. */
void ora_render_hits(const ora_model* m, ora_data* d, int cam, int width, int height, int* hit_out, double* dist_out) {
  const double* cp = d->cam_xpos + 3 * cam;
  const double* cm = d->cam_xmat + 9 * cam;
  double t = tan(0.5 * m->cam_fovy[cam] * ORA_PI / 180.0), aspect = (double)width / (double)height;
  for (int r = 0; r < height; r++)
    for (int c = 0; c < width; c++) {
      double lx = (2.0 * (c + 0.5) / width - 1.0) * t * aspect, ly = (2.0 * (r + 0.5) / height - 1.0) * t;
      double loc[3] = {lx, ly, -1.0}, vec[3];
      m3_mulv(vec, cm, loc);
      v3_normalize(vec);
      double best = -1.0;
      int hit = -1;
      for (int g = 0; g < m->ngeom; g++) {
        if (m->geom_rgba[4 * g + 3] == 0) continue;
        double x = ora_ray_geom(m->geom_type[g], d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, m->geom_size + 3 * g, cp, vec);
        if (x >= 0 && (best < 0 || x < best)) { best = x; hit = g; }
      }
      hit_out[(size_t)r * width + c] = hit;
      dist_out[(size_t)r * width + c] = best;
    }
}

/* One pixel's shadow rays, for the test harness: out[4 * li + 0..3] = n.L at the pixel's surface point for light li, the
 * first occluding geom found (-1: none), its distance along the shadow ray, the light's distance (0: directional). */
void ora_shadow_probe(const ora_model* m, ora_data* d, int cam, int width, int height, int r, int c, double* out) {
  const double* cp = d->cam_xpos + 3 * cam;
  const double* cm = d->cam_xmat + 9 * cam;
  double t = tan(0.5 * m->cam_fovy[cam] * ORA_PI / 180.0), aspect = (double)width / (double)height;
  double lx = (2.0 * (c + 0.5) / width - 1.0) * t * aspect, ly = (2.0 * (r + 0.5) / height - 1.0) * t;
  double loc[3] = {lx, ly, -1.0}, vec[3];
  m3_mulv(vec, cm, loc);
  v3_normalize(vec);
  double best = -1.0;
  int hit = -1;
  for (int g = 0; g < m->ngeom; g++) {
    if (m->geom_rgba[4 * g + 3] == 0) continue;
    double x = ora_ray_geom(m->geom_type[g], d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, m->geom_size + 3 * g, cp, vec);
    if (x >= 0 && (best < 0 || x < best)) { best = x; hit = g; }
  }
  for (int li = 0; li < m->nlight; li++) { out[4 * li] = 0; out[4 * li + 1] = -1; out[4 * li + 2] = 0; out[4 * li + 3] = 0; }
  if (hit < 0) return;
  double p[3], n[3];
  v3_addscl(p, cp, vec, best);
  ora_geom_normal(m->geom_type[hit], d->geom_xpos + 3 * hit, d->geom_xmat + 9 * hit, m->geom_size + 3 * hit, p, n);
  for (int li = 0; li < m->nlight; li++) {
    const int b = m->light_bodyid[li];
    double pos[3], dir[3], L[3], light_dist = 0;
    m3_mulv(pos, d->xmat + 9 * b, m->light_pos + 3 * li);
    v3_add(pos, pos, d->xpos + 3 * b);
    m3_mulv(dir, d->xmat + 9 * b, m->light_dir + 3 * li);
    if (m->light_directional[li]) { L[0] = -dir[0]; L[1] = -dir[1]; L[2] = -dir[2]; }
    else {
      v3_sub(L, pos, p);
      light_dist = sqrt(v3_dot(L, L));
      if (light_dist < ORA_MINVAL) continue;
      L[0] /= light_dist; L[1] /= light_dist; L[2] /= light_dist;
    }
    out[4 * li] = v3_dot(n, L);
    out[4 * li + 3] = light_dist;
    for (int g = 0; g < m->ngeom; g++) {
      if (g == hit || m->geom_rgba[4 * g + 3] == 0) continue;
      double x = ora_ray_geom(m->geom_type[g], d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, m->geom_size + 3 * g, p, L);
      if (x >= 0 && (m->light_directional[li] || x < light_dist)) { out[4 * li + 1] = g; out[4 * li + 2] = x; break; }
    }
  }
}

/* ------------------------------------------------------------------ accessors for the test harness */
typedef struct { const char* name; size_t offset; } field_t;
#define F(name) {#name, offsetof(ora_data, name)}
#include <stddef.h>
static const field_t k_fields[] = {
    F(qpos), F(qvel), F(ctrl), F(qacc_warmstart), F(xpos), F(xquat), F(xmat), F(xipos), F(ximat), F(xanchor),
    F(xaxis), F(geom_xpos), F(geom_xmat), F(site_xpos), F(site_xmat), F(cam_xpos), F(cam_xmat), F(subtree_com),
    F(cinert), F(crb), F(cdof), F(cdof_dot), F(cvel), F(cacc), F(cfrc), F(qM), F(qLD), F(qLDiagInv), F(qMdense),
    F(qfrc_bias), F(qfrc_passive), F(qfrc_actuator), F(qfrc_smooth), F(qacc_smooth), F(qfrc_constraint), F(qacc),
    F(sensordata), F(efc_J), F(efc_pos), F(efc_margin), F(efc_diagApprox), F(efc_R), F(efc_D), F(efc_vel),
    F(efc_aref), F(efc_b), F(efc_force), F(efc_AR), F(efc_KBIP)};

double* ora_field(ora_data* d, const char* name) {
  for (size_t i = 0; i < sizeof(k_fields) / sizeof(k_fields[0]); i++)
    if (strcmp(k_fields[i].name, name) == 0) return *(double**)((char*)d + k_fields[i].offset);
  return NULL;
}
int ora_ncon(const ora_data* d) { return d->ncon; }
int ora_nefc(const ora_data* d) { return d->nefc; }
int ora_niter(const ora_data* d) { return d->solver_niter; }
int ora_warnings(const ora_data* d) { return d->warn_con | (d->warn_efc << 1); }
double ora_time(const ora_data* d) { return d->time; }
/* contact i -> out[0]=dist, [1..3]=pos, [4..12]=frame, [13]=includemargin, [14]=geom1, [15]=geom2, [16]=efc_address, [17]=normal force */
void ora_contact_get(const ora_data* d, int i, double* out) {
  const ora_contact* c = d->contact + i;
  out[0] = c->dist;
  memcpy(out + 1, c->pos, 3 * sizeof(double));
  memcpy(out + 4, c->frame, 9 * sizeof(double));
  out[13] = c->includemargin; out[14] = c->geom1; out[15] = c->geom2; out[16] = c->efc_address;
  out[17] = contact_normal_force(d, c);
}
int ora_model_size(const ora_model* m, const char* name) {
#define X(field) if (strcmp(name, #field) == 0) return m->field;
  ORA_SIZE_FIELDS(X)
#undef X
  return -1;
}
