// The rigid-body forward step for ONE env copy, executed by ONE 64-lane wavefront.
//
// Replaces, per env copy, the reference's per-step host loop
//   MuJoCoParent.apply_action  (mujoco_parent.py:316-339)  scatter actions, skip_frames x mj_step
//   mj.mj_step                 (mujoco_parent.py:335; third-party mujoco==2.3.3)
//   MuJoCoParent.get_observations (mujoco_parent.py:380-392) gather sensordata|qpos|qvel per agent
// with all intermediate state (body frames, spatial inertias, mass matrix and its L'DL factor,
// contacts, constraint rows) held in LDS; HBM sees only state-in / state-out.
//
// Lane mappings change from stage to stage (lane = body, joint, dof, geom, candidate pair item,
// constraint row); stages exchange data through LDS and are separated by wv::sync().
// Arithmetic follows MuJoCo's documented pipeline; where a sum's order matters for reproducing the
// CPU oracle bit-for-bit it is noted at the loop.
#ifndef MJRL_STEP_H
#define MJRL_STEP_H

#include "mjrl_collide.h"
#include "mjrl_math.h"
#include "mjrl_model.h"
#include "mjrl_wave.h"

namespace mj {

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { SENS_TOUCH = 0, SENS_ACCELEROMETER = 1, SENS_RANGEFINDER = 2, SENS_FRAMEXAXIS = 3 };

// per-contact record in LDS (doubles)
enum { CON_DIST = 0, CON_POS = 1, CON_FRAME = 4, CON_INCL = 13, CON_MU = 14, CON_STRIDE = 16 };
// per-row record in LDS (doubles)
enum { ROW_R = 0, ROW_AREF = 1, ROW_B = 2, ROW_F = 3, ROW_ARII = 4, ROW_POS = 5, ROW_MARGIN = 6, ROW_STRIDE = 8 };
// integer header of the int region
enum { I_NCON = 0, I_NEFC = 1, I_NLIM = 2, I_NITER = 3, I_WARN = 4, I_NCAND = 5, I_HEAD = 8 };
enum { CAND_MAX = 256, MAX_DOF_DEPTH = 8 };

// LDS layout of one env copy, offsets in doubles from the env's base
struct Lay {
  int qpos, qvel, ctrl, warm, xpos, xquat, xanchor, xaxis, com, cinert, crb, cdof, cdofdot, cvel, cacc, M, LD, Dinv,
      gpos, gmat, bias, smooth, qaccs, x, qfc, qacc, con, J, row, sens, ints, total;
  int ldj;                                   // row stride of J (odd -> conflict-free column walks)
  int i_cand, i_cong1, i_cong2, i_conadr, i_rowid;   // offsets inside the int region (in ints)
};

__host__ __device__ inline void make_layout(const DevModel& m, Lay& l) {
  int o = 0;
#define REG(name, n) l.name = o; o += (n);
  REG(qpos, m.nq) REG(qvel, m.nv) REG(ctrl, m.nu) REG(warm, m.nv)
  REG(xpos, 3 * m.nbody) REG(xquat, 4 * m.nbody) REG(xanchor, 3 * m.njnt) REG(xaxis, 3 * m.njnt)
  REG(com, 3 * (m.ntree + 1)) REG(cinert, 10 * m.nbody) REG(crb, 10 * m.nbody)
  REG(cdof, 6 * m.nv) REG(cdofdot, 6 * m.nv) REG(cvel, 6 * m.nbody) REG(cacc, 6 * m.nbody)
  REG(M, m.nM) REG(LD, m.nM) REG(Dinv, m.nv)
  REG(gpos, 3 * m.ngeom) REG(gmat, 9 * m.ngeom)
  REG(bias, m.nv) REG(smooth, m.nv) REG(qaccs, m.nv) REG(x, m.nv) REG(qfc, m.nv) REG(qacc, m.nv)
  REG(con, CON_STRIDE * m.nconmax)
  l.ldj = (m.nv | 1);
  REG(J, l.ldj * m.njmax) REG(row, ROW_STRIDE * m.njmax) REG(sens, m.nsensordata + 1)
  int ni = I_HEAD;
  l.i_cand = ni; ni += CAND_MAX;
  l.i_cong1 = ni; ni += m.nconmax;
  l.i_cong2 = ni; ni += m.nconmax;
  l.i_conadr = ni; ni += m.nconmax;
  l.i_rowid = ni; ni += m.njmax;
  REG(ints, (ni + 1) / 2)
#undef REG
  l.total = o;
}

// Arguments of one step call, shared by every env copy
struct StepArgs {
  // state in HBM, [n_env][n] row-major
  real *qpos, *qvel, *ctrl, *warm, *sensordata;
  int* timestep;
  // action scatter (mujoco_parent.py:323-332): action slot -> ctrl index (mode 0) or qvel index (mode 1)
  const real* actions;       // [n_env][n_agent][act_dim], may be null (no scatter)
  const int32_t* scatter;    // [n_agent][act_dim], -1 = slot not routed to the physics
  int n_agent, act_dim, scatter_mode;
  // observation gather (mujoco_parent.py:380-392): code = kind<<24 | index; kind 0 sensordata, 1 qpos, 2 qvel
  const int32_t* gather;     // [n_agent][obs_dim], -1 = unused slot (written as 0)
  int obs_dim;
  real* obs;                 // [n_env][n_agent][obs_dim], may be null
  real* reward;              // [n_env][n_agent], may be null
  unsigned char *term, *trunc;   // [n_env][n_agent], may be null
  int max_steps, skip_frames, n_env;
  // debug dump of the env's whole LDS image after the forward pass of the last substep
  real* dbg;                 // [n_env][lay.total], may be null
  int dbg_stage;             // 0: end of forward pass; 1: right after the constraint rows are built
  int forward_only;          // 1: mj_forward semantics -- no integration, no counters (reset observations, queries)
};

#define MJ_FOR(i, n) for (int i = L; i < (n); i += 64)

// child -> parent accumulation of `n` numbers per body (row stride `stride`), deepest level first; every parent
// adds its children in descending id, the order a backward sweep over bodies meets them.
__device__ inline void tree_accumulate(const DevModel& m, real* buf, int stride, int n, bool include_world, int L) {
  for (int lev = m.maxdepth; lev >= 1; lev--) {
    MJ_FOR(b, m.nbody) {
      if (m.body_depth[b] == lev - 1 && m.body_childnum[b] > 0 && (b > 0 || include_world)) {
        int adr = m.body_childadr[b], num = m.body_childnum[b];
        for (int c = 0; c < num; c++) {
          int child = m.body_childid[adr + c];
          for (int k = 0; k < n; k++) buf[b * stride + k] += buf[child * stride + k];
        }
      }
    }
    wv::sync();
  }
}

// ------------------------------------------------------------------ position stage
__device__ inline void stage_kinematics(const DevModel& m, const Lay& l, real* S, int L) {
  if (L == 0) {
    st3(S + l.xpos, v3(0, 0, 0));
    Quat q; q.w = 1; q.x = q.y = q.z = 0;
    stq(S + l.xquat, q);
  }
  wv::sync();
  for (int lev = 1; lev <= m.maxdepth; lev++) {
    MJ_FOR(b, m.nbody) {
      if (m.body_depth[b] != lev) continue;
      int p = m.body_parentid[b];
      Quat pq = ldq(S + l.xquat + 4 * p);
      V3 pos = ld3(S + l.xpos + 3 * p) + rot(pq, ld3(m.body_pos + 3 * b));
      Quat quat = qmul(pq, ldq(m.body_quat + 4 * b));
      for (int k = 0; k < m.body_jntnum[b]; k++) {
        int j = m.body_jntadr[b] + k, qa = m.jnt_qposadr[j];
        V3 jaxis = ld3(m.jnt_axis + 3 * j), jpos = ld3(m.jnt_pos + 3 * j);
        if (m.jnt_type[j] == JNT_FREE) {
          pos = ld3(S + l.qpos + qa);
          quat = qnormalized(ldq(S + l.qpos + qa + 3));
          st3(S + l.xanchor + 3 * j, pos);
          st3(S + l.xaxis + 3 * j, rot(quat, jaxis));
        } else {
          V3 anchor = pos + rot(quat, jpos);
          V3 axis = rot(quat, jaxis);
          st3(S + l.xanchor + 3 * j, anchor);
          st3(S + l.xaxis + 3 * j, axis);
          real q = S[l.qpos + qa] - m.qpos0[qa];
          if (m.jnt_type[j] == JNT_HINGE) {
            quat = qmul(quat, axis_angle(jaxis, q));
            pos = anchor - rot(quat, jpos);
          } else {
            pos = pos + axis * q;
          }
        }
      }
      quat = qnormalized(quat);
      st3(S + l.xpos + 3 * b, pos);
      stq(S + l.xquat + 4 * b, quat);
    }
    wv::sync();
  }
}

__device__ inline V3 body_xipos(const DevModel& m, const Lay& l, const real* S, int b) {
  return ld3(S + l.xpos + 3 * b) + rot(ldq(S + l.xquat + 4 * b), ld3(m.body_ipos + 3 * b));
}

__device__ inline void stage_com_inertia(const DevModel& m, const Lay& l, real* S, int L) {
  // mass-weighted centres, accumulated up the tree exactly like a backward body sweep would
  MJ_FOR(b, m.nbody) {
    V3 xi = b ? body_xipos(m, l, S, b) : v3(0, 0, 0);
    st3(S + l.crb + 10 * b, xi * m.body_mass[b]);
  }
  wv::sync();
  tree_accumulate(m, S + l.crb, 10, 3, true, L);
  MJ_FOR(t, m.ntree) {
    int root = m.tree_rootbody[t];
    real mass = m.body_subtreemass[root];
    V3 c = mass < MJ_MINVAL ? body_xipos(m, l, S, root) : ld3(S + l.crb + 10 * root) * (1.0 / mass);
    st3(S + l.com + 3 * t, c);
  }
  wv::sync();
  // body inertias about their tree's centre of mass
  MJ_FOR(b, m.nbody) {
    real ci[10];
    int t = m.body_treeid[b];
    if (b == 0 || t < 0) {
      for (int k = 0; k < 10; k++) ci[k] = 0;
    } else {
      Quat q = ldq(S + l.xquat + 4 * b);
      V3 xi = ld3(S + l.xpos + 3 * b) + rot(q, ld3(m.body_ipos + 3 * b));
      M3 ximat = qmat(qmul(q, ldq(m.body_iquat + 4 * b)));
      inert_com(ci, m.body_inertia + 3 * b, ximat, xi - ld3(S + l.com + 3 * t), m.body_mass[b]);
    }
    for (int k = 0; k < 10; k++) { S[l.cinert + 10 * b + k] = ci[k]; S[l.crb + 10 * b + k] = ci[k]; }
  }
  // joint motion axes in the same frame
  MJ_FOR(j, m.njnt) {
    int b = m.jnt_bodyid[j], da = m.jnt_dofadr[j];
    V3 off = ld3(S + l.com + 3 * m.body_treeid[b]) - ld3(S + l.xanchor + 3 * j);
    if (m.jnt_type[j] == JNT_FREE) {
      M3 xm = qmat(ldq(S + l.xquat + 4 * b));
      for (int k = 0; k < 3; k++) {
        real* c = S + l.cdof + 6 * (da + k);
        for (int r = 0; r < 6; r++) c[r] = (r == 3 + k) ? 1.0 : 0.0;
      }
      for (int k = 0; k < 3; k++) {
        real* c = S + l.cdof + 6 * (da + 3 + k);
        V3 axis = col(xm, k);
        st3(c, axis);
        st3(c + 3, cross(axis, off));
      }
    } else {
      real* c = S + l.cdof + 6 * da;
      V3 axis = ld3(S + l.xaxis + 3 * j);
      if (m.jnt_type[j] == JNT_HINGE) { st3(c, axis); st3(c + 3, cross(axis, off)); }
      else { st3(c, v3(0, 0, 0)); st3(c + 3, axis); }
    }
  }
  wv::sync();
}

__device__ inline void stage_crb(const DevModel& m, const Lay& l, real* S, int L) {
  tree_accumulate(m, S + l.crb, 10, 10, false, L);
  MJ_FOR(i, m.nv) {
    real buf[6], c[6], crb[10];
    int b = m.dof_bodyid[i];
    for (int k = 0; k < 10; k++) crb[k] = S[l.crb + 10 * b + k];
    for (int k = 0; k < 6; k++) c[k] = S[l.cdof + 6 * i + k];
    inert_mul(buf, crb, c);
    int adr = m.dof_Madr[i];
    bool diag = true;
    for (int j = i; j >= 0; j = m.dof_parentid[j]) {
      real cj[6];
      for (int k = 0; k < 6; k++) cj[k] = S[l.cdof + 6 * j + k];
      real v = dot6(cj, buf);
      if (diag) { v = m.dof_armature[i] + v; diag = false; }
      S[l.M + adr] = v;
      S[l.LD + adr] = v;
      adr++;
    }
  }
  wv::sync();
}

// in-place sparse L'DL of the matrix at S[ld..] (dof_Madr layout), Featherstone order; dinv gets 1/D
__device__ inline void factor_ld(const DevModel& m, real* S, int ld, int dinv, int L) {
  for (int k = m.nv - 1; k >= 0; k--) {
    int D = m.dof_depth[k], kk = m.dof_Madr[k];
    int a = L / MAX_DOF_DEPTH, t = L % MAX_DOF_DEPTH;
    bool valid = a < D && t < D - a;
    real tmp = 0, val = 0;
    int ij = 0, ki = kk + 1 + a;
    if (valid) {
      int i = m.M_colid[ki];
      tmp = S[ld + ki] / S[ld + kk];
      ij = m.dof_Madr[i];
      val = S[ld + ij + t] - tmp * S[ld + ki + t];
    }
    wv::sync();
    if (valid) {
      S[ld + ij + t] = val;
      if (t == 0) S[ld + ki] = tmp;
    }
    if (L == 0) S[dinv + k] = 1.0 / S[ld + kk];
    wv::sync();
  }
}

// x <- M^-1 x with the factor at ld/dinv; x lives in LDS at S[x..x+nv).  Level-parallel over dof depth; every
// dof applies its terms in the order the serial sweeps would (descendants descending, ancestors walking up).
__device__ inline void solve_ld(const DevModel& m, real* S, int ld, int dinv, int x, int L, bool backward, bool scale,
                                bool forward) {
  if (backward) {
    for (int lev = m.maxdofdepth - 1; lev >= 0; lev--) {
      MJ_FOR(i, m.nv) {
        if (m.dof_depth[i] != lev) continue;
        int adr = m.dof_descadr[i], num = m.dof_descnum[i];
        real xi = S[x + i];
        for (int c = num - 1; c >= 0; c--) {
          int e = m.desc_Madr[adr + c];
          xi -= S[ld + e] * S[x + m.M_rowid[e]];
        }
        S[x + i] = xi;
      }
      wv::sync();
    }
  }
  if (scale) {
    MJ_FOR(i, m.nv) S[x + i] *= S[dinv + i];
    wv::sync();
  }
  if (forward) {
    for (int lev = 1; lev <= m.maxdofdepth; lev++) {
      MJ_FOR(i, m.nv) {
        if (m.dof_depth[i] != lev) continue;
        int adr = m.dof_Madr[i] + 1;
        real xi = S[x + i];
        for (int j = m.dof_parentid[i]; j >= 0; j = m.dof_parentid[j]) xi -= S[ld + adr++] * S[x + j];
        S[x + i] = xi;
      }
      wv::sync();
    }
  }
}

__device__ inline void stage_geoms(const DevModel& m, const Lay& l, real* S, int L) {
  MJ_FOR(g, m.ngeom) {
    int b = m.geom_bodyid[g];
    Quat bq = ldq(S + l.xquat + 4 * b);
    st3(S + l.gpos + 3 * g, ld3(S + l.xpos + 3 * b) + rot(bq, ld3(m.geom_pos + 3 * g)));
    M3 gm = qmat(qmul(bq, ldq(m.geom_quat + 4 * g)));
    for (int k = 0; k < 9; k++) S[l.gmat + 9 * g + k] = gm.m[k];
  }
  wv::sync();
}

// ------------------------------------------------------------------ collision
__device__ inline void stage_collision(const DevModel& m, const Lay& l, real* S, int L) {
  int* I = (int*)(S + l.ints);
  int ncand = 0, warn = 0;
  // broad phase: bounding spheres (planes: signed distance of the other geom's bounding sphere)
  for (int base = 0; base < m.npair; base += 64) {
    int p = base + L;
    bool pass = false;
    if (p < m.npair) {
      int g1 = m.pair_geom[2 * p], g2 = m.pair_geom[2 * p + 1];
      real margin = fmax(m.geom_margin[g1], m.geom_margin[g2]);
      V3 dif = ld3(S + l.gpos + 3 * g2) - ld3(S + l.gpos + 3 * g1);
      if (m.geom_type[g1] == GEOM_PLANE) {
        V3 n = v3(S[l.gmat + 9 * g1 + 2], S[l.gmat + 9 * g1 + 5], S[l.gmat + 9 * g1 + 8]);
        pass = !(dot(dif, n) > m.geom_rbound[g2] + margin);
      } else {
        real bound = m.geom_rbound[g1] + m.geom_rbound[g2] + margin;
        pass = !(dot(dif, dif) > bound * bound);
      }
    }
    unsigned long long mask = wv::ballot(pass);
    if (pass) {
      int slot = ncand + wv::popc(mask & ((1ull << L) - 1ull));
      if (slot < CAND_MAX) I[l.i_cand + slot] = p;
    }
    ncand += wv::popc(mask);
  }
  if (ncand > CAND_MAX) { ncand = CAND_MAX; warn |= 4; }
  wv::sync();
  // narrow phase: lane = (candidate, item)
  int kmax = m.pair_kmax;   // items per candidate pair this model can need (set by the host: 1, 2, 4 or 8)
  int per = 64 / kmax, ncon = 0;
  for (int base = 0; base < ncand; base += per) {
    int c = base + L / kmax, k = L % kmax;
    bool hit = false;
    RawCon rc;
    int g1 = 0, g2 = 0;
    real margin = 0, gap = 0;
    if (c < ncand) {
      int p = I[l.i_cand + c];
      g1 = m.pair_geom[2 * p]; g2 = m.pair_geom[2 * p + 1];
      int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
      if (k < pair_items(t1, t2)) {
        margin = fmax(m.geom_margin[g1], m.geom_margin[g2]);
        gap = fmax(m.geom_gap[g1], m.geom_gap[g2]);
        hit = collide_item(t1, t2, ld3(S + l.gpos + 3 * g1), ldm(S + l.gmat + 9 * g1), ld3(m.geom_size + 3 * g1),
                           ld3(S + l.gpos + 3 * g2), ldm(S + l.gmat + 9 * g2), ld3(m.geom_size + 3 * g2), margin, k, rc);
      }
    }
    unsigned long long mask = wv::ballot(hit);
    if (hit) {
      int slot = ncon + wv::popc(mask & ((1ull << L) - 1ull));
      if (slot < m.nconmax) {
        real* C = S + l.con + CON_STRIDE * slot;
        C[CON_DIST] = rc.dist;
        st3(C + CON_POS, rc.pos);
        make_frame(rc.n, rc.t, C + CON_FRAME);
        C[CON_INCL] = margin - gap;
        C[CON_MU] = fmax(m.geom_friction[3 * g1], m.geom_friction[3 * g2]);
        I[l.i_cong1 + slot] = g1;
        I[l.i_cong2 + slot] = g2;
      }
    }
    ncon += wv::popc(mask);
  }
  if (ncon > m.nconmax) { ncon = m.nconmax; warn |= 1; }
  if (L == 0) { I[I_NCON] = ncon; I[I_WARN] = warn; I[I_NCAND] = ncand; }
  wv::sync();
}

// ------------------------------------------------------------------ velocity stage (comVel + RNE bias)
__device__ inline void stage_velocity(const DevModel& m, const Lay& l, real* S, int L, bool with_acc) {
  // with_acc == false: cvel, cdof_dot, cacc (bias accelerations), cfrc -> qfrc_bias
  // with_acc == true : cacc including cdof*qacc only (for the accelerometer), nothing else is touched
  if (L == 0) {
    for (int r = 0; r < 6; r++) { if (!with_acc) S[l.cvel + r] = 0; }
    S[l.cacc + 0] = S[l.cacc + 1] = S[l.cacc + 2] = 0;
    S[l.cacc + 3] = -m.gravity_x; S[l.cacc + 4] = -m.gravity_y; S[l.cacc + 5] = -m.gravity_z;
  }
  wv::sync();
  for (int lev = 1; lev <= m.maxdepth; lev++) {
    MJ_FOR(b, m.nbody) {
      if (m.body_depth[b] != lev) continue;
      int p = m.body_parentid[b];
      real cvel[6], cacc[6];
      for (int r = 0; r < 6; r++) { cvel[r] = S[l.cvel + 6 * p + r]; cacc[r] = S[l.cacc + 6 * p + r]; }
      int da = m.body_dofadr[b];
      if (!with_acc) {
        for (int k = 0; k < m.body_jntnum[b]; k++) {
          int j = m.body_jntadr[b] + k;
          if (m.jnt_type[j] == JNT_FREE) {
            for (int t = 0; t < 3; t++) {
              for (int r = 0; r < 6; r++) {
                S[l.cdofdot + 6 * (da + t) + r] = 0;
                cvel[r] += S[l.cdof + 6 * (da + t) + r] * S[l.qvel + da + t];
              }
            }
            da += 3;
            for (int t = 0; t < 3; t++) {
              real cd[6], dd[6];
              for (int r = 0; r < 6; r++) cd[r] = S[l.cdof + 6 * (da + t) + r];
              cross_motion(dd, cvel, cd);
              for (int r = 0; r < 6; r++) S[l.cdofdot + 6 * (da + t) + r] = dd[r];
            }
            for (int t = 0; t < 3; t++)
              for (int r = 0; r < 6; r++) cvel[r] += S[l.cdof + 6 * (da + t) + r] * S[l.qvel + da + t];
            da += 3;
          } else {
            real cd[6], dd[6];
            for (int r = 0; r < 6; r++) cd[r] = S[l.cdof + 6 * da + r];
            cross_motion(dd, cvel, cd);
            for (int r = 0; r < 6; r++) {
              S[l.cdofdot + 6 * da + r] = dd[r];
              cvel[r] += cd[r] * S[l.qvel + da];
            }
            da++;
          }
        }
        for (int r = 0; r < 6; r++) S[l.cvel + 6 * b + r] = cvel[r];
      }
      da = m.body_dofadr[b];
      for (int t = 0; t < m.body_dofnum[b]; t++)
        for (int r = 0; r < 6; r++) {
          cacc[r] += S[l.cdofdot + 6 * (da + t) + r] * S[l.qvel + da + t];
          if (with_acc) cacc[r] += S[l.cdof + 6 * (da + t) + r] * S[l.qacc + da + t];
        }
      for (int r = 0; r < 6; r++) S[l.cacc + 6 * b + r] = cacc[r];
      if (!with_acc) {
        // body force  I*a + v x* (I*v)   (crb's rows are free again: reuse them for cfrc)
        real ci[10], t1[6], t2[6], t3[6];
        for (int k = 0; k < 10; k++) ci[k] = S[l.cinert + 10 * b + k];
        inert_mul(t1, ci, cacc);
        inert_mul(t2, ci, cvel);
        cross_force(t3, cvel, t2);
        for (int r = 0; r < 6; r++) S[l.crb + 10 * b + r] = t1[r] + t3[r];
      }
    }
    wv::sync();
  }
  if (with_acc) return;
  if (L == 0) for (int r = 0; r < 6; r++) S[l.crb + r] = 0;
  wv::sync();
  tree_accumulate(m, S + l.crb, 10, 6, false, L);
  MJ_FOR(i, m.nv) {
    real c[6], f[6];
    int b = m.dof_bodyid[i];
    for (int r = 0; r < 6; r++) { c[r] = S[l.cdof + 6 * i + r]; f[r] = S[l.crb + 10 * b + r]; }
    S[l.bias + i] = dot6(c, f);
  }
  wv::sync();
}

// qfrc_smooth = passive - bias + actuator ; qacc_smooth = M^-1 qfrc_smooth
__device__ inline void stage_smooth(const DevModel& m, const Lay& l, real* S, int L) {
  MJ_FOR(i, m.nv) {
    real act = 0;
    for (int u = 0; u < m.nu; u++) {
      if (m.act_dofid[u] != i) continue;
      real c = S[l.ctrl + u];
      if (m.act_ctrllimited[u]) c = fmin(fmax(c, m.act_ctrlrange[2 * u]), m.act_ctrlrange[2 * u + 1]);
      act += m.act_gear[u] * c;
    }
    real passive = -m.dof_damping[i] * S[l.qvel + i];
    real sm = passive - S[l.bias + i] + act;
    S[l.smooth + i] = sm;
    S[l.qaccs + i] = sm;
  }
  wv::sync();
  solve_ld(m, S, l.LD, l.Dinv, l.qaccs, L, true, true, true);
}

// ------------------------------------------------------------------ constraint rows
__device__ inline void stage_rows(const DevModel& m, const Lay& l, real* S, int L) {
  int* I = (int*)(S + l.ints);
  int ncon = I[I_NCON], warn = I[I_WARN];
  // joint limits: item = (joint, side), lower side first
  int nlim = 0;
  for (int base = 0; base < 2 * m.njnt; base += 64) {
    int it = base + L, j = it >> 1, side = (it & 1) ? 1 : -1;
    bool active = false;
    real dist = 0;
    if (it < 2 * m.njnt && m.jnt_limited[j] && (m.jnt_type[j] == JNT_HINGE || m.jnt_type[j] == JNT_SLIDE)) {
      real value = S[l.qpos + m.jnt_qposadr[j]];
      dist = side * (m.jnt_range[2 * j + (side + 1) / 2] - value);
      active = dist < m.jnt_margin[j];
    }
    unsigned long long mask = wv::ballot(active);
    if (active) {
      int r = nlim + wv::popc(mask & ((1ull << L) - 1ull));
      if (r < m.njmax) {
        I[l.i_rowid + r] = -(it + 1);          // negative: limit row, item id it
        S[l.row + ROW_STRIDE * r + ROW_POS] = dist;
      }
    }
    nlim += wv::popc(mask);
  }
  if (nlim > m.njmax) { nlim = m.njmax; warn |= 2; }
  wv::sync();
  // contacts: address of each pyramid (serial prefix over at most nconmax entries)
  if (L == 0) {
    int adr = nlim;
    for (int c = 0; c < ncon; c++) {
      const real* C = S + l.con + CON_STRIDE * c;
      int g1 = I[l.i_cong1 + c], g2 = I[l.i_cong2 + c];
      int dim = m.geom_condim[g1] > m.geom_condim[g2] ? m.geom_condim[g1] : m.geom_condim[g2];
      int rows = dim == 1 ? 1 : 2 * ((dim < 3 ? dim : 3) - 1);
      int a = -1;
      if (C[CON_DIST] < C[CON_INCL]) {
        if (adr + rows <= m.njmax) {
          a = adr;
          for (int s = 0; s < rows; s++) I[l.i_rowid + adr + s] = c * 8 + s;
          adr += rows;
        } else warn |= 2;
      }
      I[l.i_conadr + c] = a;
    }
    I[I_NEFC] = adr; I[I_NLIM] = nlim; I[I_WARN] = warn;
  }
  wv::sync();
  int nefc = I[I_NEFC];
  // one lane per row: Jacobian row, reference acceleration, regularisation
  MJ_FOR(r, nefc) {
    real* Jr = S + l.J + l.ldj * r;
    real* R = S + l.row + ROW_STRIDE * r;
    for (int k = 0; k < m.nv; k++) Jr[k] = 0;
    int id = I[l.i_rowid + r];
    real pos, margin, diag, mu0 = 0;
    const real *solref, *solimp;
    real sref[2], simp[5];
    bool contact = id >= 0;
    if (!contact) {
      int it = -id - 1, j = it >> 1, side = (it & 1) ? 1 : -1;
      int dof = m.jnt_dofadr[j];
      Jr[dof] = -side;
      pos = R[ROW_POS];
      margin = m.jnt_margin[j];
      diag = m.dof_invweight0[dof];
      solref = m.jnt_solref + 2 * j;
      solimp = m.jnt_solimp + 5 * j;
    } else {
      int c = id >> 3, sub = id & 7;
      const real* C = S + l.con + CON_STRIDE * c;
      int g1 = I[l.i_cong1 + c], g2 = I[l.i_cong2 + c];
      int b1 = m.geom_bodyid[g1], b2 = m.geom_bodyid[g2];
      int dim = m.geom_condim[g1] > m.geom_condim[g2] ? m.geom_condim[g1] : m.geom_condim[g2];
      V3 cp = ld3(C + CON_POS), n = ld3(C + CON_FRAME);
      real mu = C[CON_MU];     // friction[0] == friction[1]: both tangent directions share it
      mu0 = mu;
      int kt = 1 + (sub >> 1);
      real sgn = (sub & 1) ? -1.0 : 1.0;
      V3 tk = ld3(C + CON_FRAME + 3 * kt);
      int t1 = m.body_treeid[b1], t2 = m.body_treeid[b2];
      V3 off1 = t1 >= 0 ? cp - ld3(S + l.com + 3 * t1) : v3(0, 0, 0);
      V3 off2 = t2 >= 0 ? cp - ld3(S + l.com + 3 * t2) : v3(0, 0, 0);
      int i1 = m.body_lastdof[b1], i2 = m.body_lastdof[b2];
      while (i1 >= 0 || i2 >= 0) {
        int i = i1 > i2 ? i1 : i2;
        V3 ca = ld3(S + l.cdof + 6 * i), cl = ld3(S + l.cdof + 6 * i + 3);
        V3 colv = v3(0, 0, 0);
        if (i2 == i) { colv = cl + cross(ca, off2); i2 = m.dof_parentid[i2]; }
        if (i1 == i) { colv = colv - (cl + cross(ca, off1)); i1 = m.dof_parentid[i1]; }
        real jn = dot(n, colv);
        Jr[i] = dim == 1 ? jn : jn + sgn * mu * dot(tk, colv);
      }
      pos = C[CON_DIST];
      margin = C[CON_INCL];
      real tran = m.body_invweight0[2 * b1] + m.body_invweight0[2 * b2];
      diag = dim == 1 ? tran : tran + mu * mu * tran;
      // solver parameters mixed by solmix
      real s1 = m.geom_solmix[g1], s2 = m.geom_solmix[g2], mix;
      if (s1 >= MJ_MINVAL && s2 >= MJ_MINVAL) mix = s1 / (s1 + s2);
      else if (s1 < MJ_MINVAL && s2 < MJ_MINVAL) mix = 0.5;
      else mix = s1 < MJ_MINVAL ? 0.0 : 1.0;
      for (int k = 0; k < 2; k++) sref[k] = mix * m.geom_solref[2 * g1 + k] + (1 - mix) * m.geom_solref[2 * g2 + k];
      for (int k = 0; k < 5; k++) simp[k] = mix * m.geom_solimp[5 * g1 + k] + (1 - mix) * m.geom_solimp[5 * g2 + k];
      solref = sref;
      solimp = simp;
      contact = dim > 1;
    }
    // velocity, smooth acceleration and warm-start acceleration along the row (ascending dof order)
    real vel = 0, ja = 0, jw = 0;
    for (int k = 0; k < m.nv; k++) {
      real jk = Jr[k];
      vel += jk * S[l.qvel + k];
      ja += jk * S[l.qaccs + k];
      jw += jk * S[l.warm + k];
    }
    real imp = impedance(solimp, pos, margin);
    real dmax = fmin(fmax(solimp[1], MJ_MINIMP), MJ_MAXIMP);
    real timeconst = fmax(solref[0], 2 * m.timestep), dampratio = solref[1];
    real K = 1.0 / fmax(MJ_MINVAL, dmax * dmax * timeconst * timeconst * dampratio * dampratio);
    real B = 2.0 / fmax(MJ_MINVAL, dmax * timeconst);
    real Rr = fmax(MJ_MINVAL, (1 - imp) / imp * diag);
    if (contact) Rr = fmax(MJ_MINVAL, 2 * mu0 * mu0 * Rr);   // pyramid edges share 2*mu^2*R(first edge)
    real aref = -B * vel - K * imp * (pos - margin);
    real jar = jw - aref;
    real Dr = 1.0 / Rr;
    R[ROW_R] = Rr;
    R[ROW_AREF] = aref;
    R[ROW_B] = ja - aref;
    R[ROW_F] = jar < 0 ? -Dr * jar : 0.0;
    R[ROW_POS] = pos;
    R[ROW_MARGIN] = margin;
  }
  wv::sync();
}

// J <- J L^-1 (row-wise back substitution against the factor), then AR_ii = sum_d B_id^2 / D_d + R_i
__device__ inline void stage_project(const DevModel& m, const Lay& l, real* S, int L) {
  int* I = (int*)(S + l.ints);
  int nefc = I[I_NEFC];
  for (int base = 0; base < nefc; base += 64) {
    int r = base + L;
    bool own = r < nefc;
    real* Jr = S + l.J + l.ldj * (own ? r : 0);
    for (int k = m.nv - 1; k >= 0; k--) {
      real v = own ? Jr[k] : 0.0;
      if (wv::ballot(v != 0.0) == 0ull) continue;
      int adr = m.dof_Madr[k] + 1;
      for (int j = m.dof_parentid[k]; j >= 0; j = m.dof_parentid[j]) {
        if (v != 0.0) Jr[j] -= v * S[l.LD + adr];
        adr++;
      }
    }
    if (own) {
      real acc = 0;
      for (int k = 0; k < m.nv; k++) acc += Jr[k] * Jr[k] * S[l.Dinv + k];
      S[l.row + ROW_STRIDE * r + ROW_ARII] = acc + S[l.row + ROW_STRIDE * r + ROW_R];
    }
  }
  wv::sync();
}

// projected Gauss-Seidel on the dual  min 1/2 f'(A+R)f + f'b, f >= 0, with A = B D^-1 B' never formed:
// lane d carries u_d = (B' f)_d, a row's residual is one wave reduction, its update one fused multiply-add.
__device__ inline void stage_pgs(const DevModel& m, const Lay& l, real* S, int L) {
  int* I = (int*)(S + l.ints);
  int nefc = I[I_NEFC];
  bool dof = L < m.nv;             // nv <= 64 (checked at create)
  real dinv = dof ? S[l.Dinv + L] : 0.0;
  real u = 0;
  if (nefc == 0) {
    if (L == 0) I[I_NITER] = 0;
    MJ_FOR(i, m.nv) { S[l.qfc + i] = 0; S[l.qacc + i] = S[l.qaccs + i]; S[l.warm + i] = S[l.qaccs + i]; }
    wv::sync();
    return;
  }
  // warm start: keep the forces implied by last step's acceleration only if they beat f = 0
  if (dof)
    for (int r = 0; r < nefc; r++) u += S[l.J + l.ldj * r + L] * S[l.row + ROW_STRIDE * r + ROW_F];
  real part = 0.5 * dinv * u * u;
  MJ_FOR(r, nefc) {
    const real* R = S + l.row + ROW_STRIDE * r;
    part += 0.5 * R[ROW_R] * R[ROW_F] * R[ROW_F] + R[ROW_F] * R[ROW_B];
  }
  real cost = wv::sum(part);
  if (cost > 0) {
    u = 0;
    MJ_FOR(r, nefc) S[l.row + ROW_STRIDE * r + ROW_F] = 0;
  }
  wv::sync();
  real scale = 1.0 / (m.meaninertia * (m.nv > 1 ? m.nv : 1));
  int iter = 0;
  while (iter < m.iterations) {
    real improvement = 0;
    for (int i = 0; i < nefc; i++) {
      const real* R = S + l.row + ROW_STRIDE * i;
      real bid = dof ? S[l.J + l.ldj * i + L] : 0.0;
      real fi = R[ROW_F], Ri = R[ROW_R], bi = R[ROW_B], aii = R[ROW_ARII];
      real res = wv::sum(bid * dinv * u) + Ri * fi + bi;
      real fn = fi - res / aii;
      if (fn < 0) fn = 0;
      real delta = fn - fi;
      real change = 0.5 * delta * delta * aii + delta * res;
      if (change > 1e-10) { fn = fi; delta = 0; change = 0; }
      improvement -= change;
      u += delta * bid;
      if (L == 0) S[l.row + ROW_STRIDE * i + ROW_F] = fn;
    }
    iter++;
    wv::sync();
    if (improvement * scale < m.tolerance) break;
  }
  if (L == 0) I[I_NITER] = iter;
  // back to joint space: qfrc_constraint = L' u ; qacc = qacc_smooth + L^-1 D^-1 u
  if (dof) S[l.x + L] = u;
  wv::sync();
  MJ_FOR(i, m.nv) {
    int adr = m.dof_descadr[i], num = m.dof_descnum[i];
    real q = S[l.x + i];
    for (int c = 0; c < num; c++) {
      int e = m.desc_Madr[adr + c];
      q += S[l.LD + e] * S[l.x + m.M_rowid[e]];
    }
    S[l.qfc + i] = q;
  }
  wv::sync();
  solve_ld(m, S, l.LD, l.Dinv, l.x, L, false, true, true);
  MJ_FOR(i, m.nv) {
    real a = S[l.qaccs + i] + S[l.x + i];
    S[l.qacc + i] = a;
    S[l.warm + i] = a;
  }
  wv::sync();
}

// ------------------------------------------------------------------ sensors (mj_forward's sensor stage)
__device__ inline void stage_sensors(const DevModel& m, const Lay& l, real* S, int L) {
  int* I = (int*)(S + l.ints);
  bool need_acc = false;
  for (int s = 0; s < m.nsensor; s++) need_acc |= (m.sensor_type[s] == SENS_ACCELEROMETER);
  if (need_acc) stage_velocity(m, l, S, L, true);
  for (int s = 0; s < m.nsensor; s++) {
    int site = m.sensor_objid[s], adr = m.sensor_adr[s], body = m.site_bodyid[site], type = m.sensor_type[s];
    real cutoff = m.sensor_cutoff[s];
    Quat bq = ldq(S + l.xquat + 4 * body);
    V3 sp = ld3(S + l.xpos + 3 * body) + rot(bq, ld3(m.site_pos + 3 * site));
    M3 sm = qmat(qmul(bq, ldq(m.site_quat + 4 * site)));
    if (type == SENS_RANGEFINDER) {
      V3 vec = col(sm, 2);
      real best = 1e300;
      MJ_FOR(g, m.ngeom) {
        if (m.geom_bodyid[g] == body || m.geom_rgba[4 * g + 3] == 0) continue;
        real x = ray_geom(m.geom_type[g], ld3(S + l.gpos + 3 * g), ldm(S + l.gmat + 9 * g), ld3(m.geom_size + 3 * g), sp, vec);
        if (x >= 0 && x < best) best = x;
      }
      best = wv::min_pos(best);
      real out = best > 1e299 ? -1.0 : best;
      if (cutoff > 0 && out > cutoff) out = cutoff;
      if (L == 0) S[l.sens + adr] = out;
    } else if (type == SENS_TOUCH) {
      real part = 0;
      int ncon = I[I_NCON];
      MJ_FOR(c, ncon) {
        int a = I[l.i_conadr + c];
        int g1 = I[l.i_cong1 + c], g2 = I[l.i_cong2 + c];
        int b1 = m.geom_bodyid[g1], b2 = m.geom_bodyid[g2];
        if (a < 0 || (b1 != body && b2 != body)) continue;
        int dim = m.geom_condim[g1] > m.geom_condim[g2] ? m.geom_condim[g1] : m.geom_condim[g2];
        int rows = dim == 1 ? 1 : 2 * ((dim < 3 ? dim : 3) - 1);
        real fn = 0;
        for (int r = 0; r < rows; r++) fn += S[l.row + ROW_STRIDE * (a + r) + ROW_F];
        if (fn <= 0) continue;
        const real* C = S + l.con + CON_STRIDE * c;
        V3 ray = ld3(C + CON_FRAME) * (b2 == body ? -1.0 : 1.0);
        if (ray_sphere_at(sp, m.site_size[3 * site], ld3(C + CON_POS), ray) >= 0) part += fn;
      }
      real out = wv::sum(part);
      if (cutoff > 0 && out > cutoff) out = cutoff;
      if (L == 0) S[l.sens + adr] = out;
    } else if (type == SENS_ACCELEROMETER) {
      if (L == 0) {
        int t = m.body_treeid[body];
        V3 off = sp - ld3(S + l.com + 3 * (t < 0 ? m.ntree : t));
        V3 wa = ld3(S + l.cacc + 6 * body), la = ld3(S + l.cacc + 6 * body + 3);
        V3 wv_ = ld3(S + l.cvel + 6 * body), lv = ld3(S + l.cvel + 6 * body + 3);
        V3 acc = la + cross(wa, off), vel = lv + cross(wv_, off);
        V3 wl = mulT(sm, wv_), vl = mulT(sm, vel), al = mulT(sm, acc);
        V3 o = al + cross(wl, vl);
        real out[3] = {o.x, o.y, o.z};
        for (int k = 0; k < 3; k++) {
          if (cutoff > 0) out[k] = fmin(fmax(out[k], -cutoff), cutoff);
          S[l.sens + adr + k] = out[k];
        }
      }
    } else {
      int c = type - SENS_FRAMEXAXIS;
      if (L == 0) st3(S + l.sens + adr, col(sm, c));
    }
  }
  wv::sync();
}

// ------------------------------------------------------------------ integrator
__device__ inline void stage_euler(const DevModel& m, const Lay& l, real* S, int L) {
  real h = m.timestep;
  bool damped = false;
  for (int i = 0; i < m.nv; i++) damped |= (m.dof_damping[i] > 0);
  if (damped) {
    // (M + h*diag(damping)) qacc = qfrc_smooth + qfrc_constraint
    MJ_FOR(e, m.nM) S[l.LD + e] = S[l.M + e];
    wv::sync();
    MJ_FOR(i, m.nv) {
      S[l.LD + m.dof_Madr[i]] += h * m.dof_damping[i];
      S[l.x + i] = S[l.smooth + i] + S[l.qfc + i];
    }
    wv::sync();
    factor_ld(m, S, l.LD, l.Dinv, L);
    solve_ld(m, S, l.LD, l.Dinv, l.x, L, true, true, true);
  } else {
    MJ_FOR(i, m.nv) S[l.x + i] = S[l.qacc + i];
    wv::sync();
  }
  MJ_FOR(i, m.nv) S[l.qvel + i] += h * S[l.x + i];
  wv::sync();
  MJ_FOR(j, m.njnt) {
    int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
    if (m.jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) S[l.qpos + qa + k] += h * S[l.qvel + da + k];
      real len;
      V3 w = normalized(ld3(S + l.qvel + da + 3), &len);
      Quat q = qmul(qnormalized(ldq(S + l.qpos + qa + 3)), axis_angle(w, h * len));
      stq(S + l.qpos + qa + 3, q);
    } else {
      S[l.qpos + qa] += h * S[l.qvel + da];
    }
  }
  wv::sync();
}

// ------------------------------------------------------------------ one env copy, one step() call
__device__ inline void env_step(const DevModel& m, const StepArgs& a, real* S) {
  const int L = wv::lane();
  const int env = wv::env_index();
  Lay l;
  make_layout(m, l);
  // state in
  MJ_FOR(i, m.nq) S[l.qpos + i] = a.qpos[(size_t)env * m.nq + i];
  MJ_FOR(i, m.nv) { S[l.qvel + i] = a.qvel[(size_t)env * m.nv + i]; S[l.warm + i] = a.warm[(size_t)env * m.nv + i]; }
  MJ_FOR(i, m.nu) S[l.ctrl + i] = a.ctrl[(size_t)env * m.nu + i];
  wv::sync();
  // scatter the physical part of every agent's action (mujoco_parent.py:323-332)
  if (a.actions) {
    MJ_FOR(it, a.n_agent * a.act_dim) {
      int idx = a.scatter[it];
      if (idx >= 0) {
        real v = a.actions[(size_t)env * a.n_agent * a.act_dim + it];
        if (a.scatter_mode == 0) S[l.ctrl + idx] = v; else S[l.qvel + idx] = v;
      }
    }
    wv::sync();
  }
  for (int frame = 0; frame < a.skip_frames; frame++) {
    stage_kinematics(m, l, S, L);
    stage_com_inertia(m, l, S, L);
    stage_crb(m, l, S, L);
    factor_ld(m, S, l.LD, l.Dinv, L);
    stage_geoms(m, l, S, L);
    stage_collision(m, l, S, L);
    stage_velocity(m, l, S, L, false);
    stage_smooth(m, l, S, L);
    stage_rows(m, l, S, L);
    if (a.dbg && a.dbg_stage == 1 && frame == a.skip_frames - 1)
      MJ_FOR(i, l.total) a.dbg[(size_t)env * l.total + i] = S[i];
    stage_project(m, l, S, L);
    stage_pgs(m, l, S, L);
    stage_sensors(m, l, S, L);
    if (a.dbg && a.dbg_stage == 0 && frame == a.skip_frames - 1)
      MJ_FOR(i, l.total) a.dbg[(size_t)env * l.total + i] = S[i];
    if (!a.forward_only) stage_euler(m, l, S, L);
  }
  // state out
  if (!a.forward_only) {
    MJ_FOR(i, m.nq) a.qpos[(size_t)env * m.nq + i] = S[l.qpos + i];
    MJ_FOR(i, m.nv) a.qvel[(size_t)env * m.nv + i] = S[l.qvel + i];
    MJ_FOR(i, m.nu) a.ctrl[(size_t)env * m.nu + i] = S[l.ctrl + i];
  }
  MJ_FOR(i, m.nv) a.warm[(size_t)env * m.nv + i] = S[l.warm + i];
  MJ_FOR(i, m.nsensordata) a.sensordata[(size_t)env * m.nsensordata + i] = S[l.sens + i];
  // per-agent observation gather: sensordata | qpos | qvel (sensordata is the pre-integration forward pass,
  // qpos/qvel are post-integration, exactly as the reference reads them after mj_step)
  if (a.obs) {
    MJ_FOR(it, a.n_agent * a.obs_dim) {
      int code = a.gather[it];
      real v = 0;
      if (code >= 0) {
        int kind = code >> 24, idx = code & 0xFFFFFF;
        v = kind == 0 ? S[l.sens + idx] : (kind == 1 ? S[l.qpos + idx] : S[l.qvel + idx]);
      }
      a.obs[(size_t)env * a.n_agent * a.obs_dim + it] = v;
    }
  }
  if (a.forward_only) return;
  // truncation is evaluated before the counter moves (mujoco_rl.py:279,288)
  int ts = a.timestep[env];
  MJ_FOR(ag, a.n_agent) {
    if (a.reward) a.reward[(size_t)env * a.n_agent + ag] = 0;
    if (a.term) a.term[(size_t)env * a.n_agent + ag] = 0;
    if (a.trunc) a.trunc[(size_t)env * a.n_agent + ag] = ts >= a.max_steps;
  }
  wv::sync();
  if (L == 0) a.timestep[env] = ts + 1;
}

}  // namespace mj

#endif
