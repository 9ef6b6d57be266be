// The rigid-body forward step for ONE env copy, executed by ONE 64-lane wavefront.
//
// Replaces, per env copy, the reference's per-step host loop
//   MuJoCoParent.apply_action  (mujoco_parent.py:316-339)  scatter actions, skip_frames x mj_step
//   mj.mj_step                 (mujoco_parent.py:335; third-party mujoco==2.3.3)
//   MuJoCoParent.get_observations (mujoco_parent.py:380-392) gather sensordata|qpos|qvel per agent
// with all intermediate state (body frames, spatial inertias, mass matrix and its L'DL factor,
// contacts, constraint rows) held in LDS; HBM sees only state-in / state-out.
//
// Lane mappings change from stage to stage (lane = body, joint, dof, geom, narrow-phase work item,
// constraint row); stages exchange data through LDS and are separated by wv::sync().  Per-lane model
// constants (the lane's body / dof record) are read from HBM once per launch into registers (LaneK).
// The model must fit one wave: nbody, njnt, nv <= 64; ngeom <= 128 (checked by mjrl_create).  Geoms past the 64th --
// arenas with more scenery; the model compiler folds jointless bodies into the world when there are too many bodies --
// take a second pass in the three places where a lane stands for a geom (geom frames, rangefinder targets, the ray
// kernel's candidates); a level with at most 64 geoms compiles to the code it had.
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
enum { CON_DIST = 0, CON_POS = 1, CON_FRAME = 4, CON_INCL = 13, CON_MU = 14, CON_STRIDE = 15 };
// per-row record in LDS (doubles); ROW_F doubles as the constraint position until the row is built
enum { ROW_R = 0, ROW_B = 1, ROW_F = 2, ROW_ARII = 3, ROW_STRIDE = 4 };
// integer header of the int region
enum { I_NCON = 0, I_NEFC = 1, I_NLIM = 2, I_NITER = 3, I_WARN = 4, I_NITEM = 5, I_COST = 6, I_HEAD = 8 };
enum { MAX_DOF_DEPTH = 8 };
// A constraint row is stored in JW doubles, in one of two forms.
// Tree-local (models in the tree-row lane map, rows inside one kinematic tree -- all rows of the solver's parallel
// path): slot i is the tree's i-th dof, i.e. the slot a lane reads is its position in its row of 16 lanes.
// Compact (rows that couple two trees; every row of a model outside the lane map): a row touches the dofs of at most
// two ancestor chains (the bodies of a contact), at most MAX_DOF_DEPTH dofs each; slots 0..7 follow the primary chain
// from its deepest dof up to the tree root, slots 8..15 the secondary chain where it is not also the primary one.
// The row's chain code says which dofs those are: bits 0-5 deepest dof of the primary chain, 6-8 its depth, 9-15 one
// plus the deepest dof of the secondary chain (0: none), 16-18 its depth.
// Bits 19-22 of a row's info word hold the kinematic tree of the row plus 2 (0: the row couples two trees).
enum { JW = 2 * MAX_DOF_DEPTH, CHAIN_BITS = 19 };

// LDS layout of one env copy, offsets in doubles from the env's base.  Its size decides how many copies a CU holds
// at once (160 KiB / size) and with that how much of the step's latency is hidden, so everything with a short life
// shares storage.  The block `u` serves two lifetimes: {cinert + a scratch area} from the kinematics to the end of
// the bias forces, {J, row} from the constraint-row build to the sensors.  The scratch area in turn holds, one after
// the other: {xanchor, xaxis} (kinematics .. joint axes), {crb} (composite inertias), {gsize, work items, geom frames}
// (geoms .. collision; the few later readers of a geom's frame recompute it from its body's, geom_frame()) and {body forces in the crb slots, cvel, cdofdot, cacc} (velocity stage .. bias forces; a model with an
// accelerometer re-reads the last three after the solve and keeps them outside `u`).
struct Lay {
  int qpos, qvel, ctrl, warm, xpos, xquat, com, cdof, cdofdot, cvel, cacc, LD, Dinv, gpos, gquat, bias, smooth, qaccs,
      x, qfc, qacc, con, sens, zero, gsize, tab, ints, u, total;
  int xanchor, xaxis, cinert, crb;   // inside u, first lifetime
  int J, row;                        // inside u, second lifetime
  int ldj;                           // row stride of J (= JW, the compact row width)
  // offsets from the int region's base, in ints.  i_item (collision work items) points into the scratch area of `u`;
  // i_rowid holds the row ids during the row build and the solver's per-tree row lists afterwards
  int i_item, i_cong1, i_cong2, i_conadr, i_rowid, i_rowinfo;
};

// structure tables in LDS, one byte per entry when every value fits: dof ids and counts (<= nv), body ids + 1 (<= nbody),
// depths, geom types, and addresses inside the sparse inertia layout (< nM); 16-bit words otherwise
__host__ __device__ inline bool tab_in_bytes(const DevModel& m) { return m.nM <= 256 && m.nv < 255 && m.nbody < 255; }
__host__ __device__ inline void make_layout(const DevModel& m, Lay& l) {
  int o = 0;
#define REG(name, n) l.name = o; o += (n);
  REG(qpos, m.nq) REG(qvel, m.nv) REG(ctrl, m.nu) REG(warm, m.nv)
  REG(xpos, 3 * m.nbody) REG(xquat, 4 * m.nbody) REG(com, 3 * (m.ntree + 1))
  REG(cdof, 6 * m.nv)
  if (m.has_accel) { REG(cvel, 6 * m.nbody) REG(cdofdot, 6 * m.nv) REG(cacc, 6 * m.nbody) }
  REG(LD, m.nM) REG(Dinv, m.nv)        // (the unfactorised inertia matrix waits for the integrator in registers: MKeep)
  REG(bias, m.nv) REG(smooth, m.nv) REG(qaccs, m.nv) REG(qfc, m.nv)
  l.x = l.bias;      // solver / integrator temporary: the bias forces are dead once qfrc_smooth exists
  l.qacc = l.warm;   // the solver leaves the new acceleration in both; the old warm start is last read by the row build
  REG(con, CON_STRIDE * m.nconmax) REG(sens, m.nsensordata + 1)
  REG(zero, 1)                   // holds 0.0: where the triangular solves have no factor entry they read this
  REG(tab, tab_in_bytes(m) ? (m.ntab + 7) / 8 : (m.ntab + 3) / 4)     // structure tables staged once per launch
  int ni = I_HEAD;
  l.i_cong1 = ni; ni += m.nconmax;
  l.i_cong2 = ni; ni += m.nconmax;
  l.i_conadr = ni; ni += m.nconmax;
  l.i_rowid = ni; ni += m.njmax;
  l.i_rowinfo = ni; ni += m.njmax;
  REG(ints, (ni + 1) / 2)
  l.u = o;
  l.cinert = o;
  const int sb = o + 10 * m.nbody;
  l.xanchor = sb; l.xaxis = sb + 3 * m.njnt;
  l.crb = sb;
  l.gsize = sb; l.i_item = 2 * (sb + 3 * m.ngeom - l.ints);
  l.gpos = sb + 3 * m.ngeom + (m.nitemmax + 1) / 2; l.gquat = l.gpos + 3 * m.ngeom;
  int scratch = 6 * m.njnt;
  if (10 * m.nbody > scratch) scratch = 10 * m.nbody;
  if (10 * m.ngeom + (m.nitemmax + 1) / 2 > scratch) scratch = 10 * m.ngeom + (m.nitemmax + 1) / 2;
  if (!m.has_accel) {
    // (the bias-force recursion keeps its per-body forces in the crb slots, so the velocities start after those)
    l.cvel = sb + 10 * m.nbody; l.cdofdot = l.cvel + 6 * m.nbody; l.cacc = l.cdofdot + 6 * m.nv;
    if (22 * m.nbody + 6 * m.nv > scratch) scratch = 22 * m.nbody + 6 * m.nv;
  }
  int first = 10 * m.nbody + scratch;
  l.ldj = JW;
  l.J = o; l.row = l.J + l.ldj * m.njmax;
  int second = l.ldj * m.njmax + ROW_STRIDE * m.njmax;
  o += first > second ? first : second;
#undef REG
  l.total = o;
}

// stage clock of a wave (diagnostic launches only): time of the last stamp, and in lane k the cycles of stage k so far
struct Stamps {
  unsigned long long prev, mine;
};

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
  int max_steps, n_env;
  int skip_frames;           // physics frames in THIS launch: 0 or 1 (the host loops over the step's skipFrames)
  int more_frames;           // 1: further launches of the same step follow -- no counters, observations or plugin ops yet
  // debug dump of the env's whole LDS image after the forward pass of the last substep
  real* dbg;                 // [n_env][lay.total], may be null
  int dbg_stage;             // 0: end of forward pass; 1: right after the constraint rows are built
  int forward_only;          // 1: mj_forward semantics -- no integration, no counters; writes sensordata and the warm
                             // start like mj_forward.  2: the same as a pure query (mjrl_query): the warm start stays
  // In-launch reset (mjrl_set_step_reset_mask): a copy whose byte is set starts this step from the reset image instead of
  // its state rows -- qpos0, zero velocity and controls, the warm start mj_forward leaves at the reset state (reset_warm,
  // computed once per model at create), step counter 0, an empty data store -- i.e. `env.reset(); env.step(a)` of the
  // reference's sampling loops (fps_benchmark.py:33-38, mujoco_rl.py:291-331) in one launch.
  // A byte of 2 is a reset WITHOUT a step: the copy goes to the reset image, no physics frame runs, the observation is
  // the reset observation (what reset() returns, mujoco_rl.py:314), reward 0, flags clear, step counter 0 -- the step a
  // Gymnasium vector env takes for a copy whose episode ended in the step before ("next-step" autoreset).
  const unsigned char* reset_mask;   // [n_env] device bytes, may be null
  int first_frame;                   // 1 in the first launch of a step (a flagged copy loads the reset image only there)
  // Autoreset kept by the kernel itself (mjrl_set_autoreset): auto_mask[e] = the copy's episode ended in its last step
  // (a termination or truncation flag of any agent), written by every step's last launch; a copy whose byte is set is
  // reset by the next step -- auto_mode 1: without a step (Gymnasium next-step autoreset), 2: reset, then step
  // (`env.reset(); env.step(a)` of the reference's sampling loops).  No host involvement, no mask to upload.
  unsigned char* auto_mask;          // [n_env], may be null
  int auto_mode;
  const real* reset_sens;            // [nsensordata] sensor readings of the reset image
  // Runge-Kutta (<option integrator="RK4">, benchmarking/levels/Ant.xml:3): a physics frame is four launches, one forward
  // pass each; rk_stage 0..3 says which, rk [n_env][nq + 3 nv] keeps the frame's start state and the weighted sums of the
  // passes' derivatives between them (x0 qpos | x0 qvel | sum b_i qvel_i | sum b_i qacc_i)
  int rk_stage;
  real* rk;
  // what the copy's last physics frame did: [n_env][4] = contacts, constraint rows, solver sweeps, cap-warning bits
  // (data.ncon, data.nefc, data.solver_iter, data.warning of MjData); may be null
  int* stats;
  const real* reset_warm;            // [nv]
  unsigned long long* stamps;   // diagnostic: per-stage wave-clock sums over all env copies [N_STAMPS], may be null
  // Fused plugin vocabulary, run after the gather in list order, agent-minor, strictly sequentially -- the order of the
  // reference's plugin loop (mujoco_rl.py:215-241 dynamics, :276-277 rewards, :281-286 dones).  Op word layout: OP_*.
  const int32_t* prog_i;       // [n_op][8]
  const real* prog_f;          // [n_op][4]
  int n_op, n_slot;
  const int32_t* agent_body;   // [n_agent] body id of each agent
  const int32_t* agent_obs_len;   // [n_agent] length of the physical part of the agent's observation
  real* store;                 // [n_env][n_agent][n_slot] device data store; NaN = key not present
  // Object tags of the level's info JSON (mujoco_rl.py:93-112, 355-378) as tables: tag t names the objects
  // tag_ref[tag_adr[t] .. + tag_num[t]), in filter_by_tag's order; an entry is kind << 16 | id (kind 0 body -> xipos,
  // 1 geom -> xpos, as get_data / distance resolve a name, mujoco_parent.py:404-416, 440-446)
  const int32_t* tag_adr;
  const int32_t* tag_num;
  const int32_t* tag_ref;
  int env_base;                // global id of copy 0 (sharded batches): random choices are keyed on the global copy id
  // Per-copy level variant (an xmlPath list whose levels differ in colours only, Testing/levels/Model2-10.xml): chosen at
  // every reset like the reference's random.choice (mujoco_parent.py:352), keyed on (seed, global copy id, episode)
  int* variant;                // [n_env], may be null
  int* episode;                // [n_env] resets so far (always kept: part of the key of every on-device random choice)
  int n_variant;
  unsigned long long variant_seed;
  // Optional copy of the forward pass's frames for host-side plugin queries (data.body().xipos, data.contact ...,
  // mujoco_parent.py:404-416, 472-475): [n_env][frame_doubles] = xpos | xquat | gpos | gquat | ncon | contact geoms.
  // The reference reads those after mj_step, i.e. as the forward pass inside the step left them (pre-integration).
  real* frames;
  // Optional scene rows for the ray caster (mjrl_set_scene_cache): [n_env][scene_doubles] = geom pos 3G | geom matrix 9G |
  // camera pos 3C | camera matrix 9C | light pos 3N | light dir 3N, as the forward pass of this launch left them --
  // what mjv_updateScene reads out of MjData after mj_step (mujoco_parent.py:533): frames one integration older than
  // qpos.  reset_scene [scene_doubles] is the row of the reset image, for copies that are reset without a physics frame.
  real* scene;
  const real* reset_scene;
  // I/O layout for single-agent consumers (mjrl_set_io_layout; the vector-env adapter): io_agent >= 0 -- the action buffer
  // holds that agent's row only, [n_env][act_dim] (the other agents' slots read 0), and only its observation row is
  // written, [n_env][obs_dim]; obs_f32 -- observations are stored as float (half the bytes a host caller pulls over
  // PCIe).  Rewards and flags stay [n_env][n_agent].  (io_agent1 = agent + 1: a zero-filled StepArgs has the layouts
  // documented at the fields above.)
  int io_agent1, obs_f32;
  // The lanes' records of the model (build_lane_records: LANE_REC_INTS ints, built once per handle by mjrl_create)
  const int32_t* lane_rec;
  // The batch leaves every SIMD at most one wave (n_env <= 4 x the CUs, mjrl_create): the solver forms that need more
  // than 256 registers cost no residency then (stage_pgs: `few`).  Read by the generic kernels; a specialised kernel
  // is built for one case or the other (-DMJRL_FEW=1, kernel_cache.code_object(..., few=True)).
  int few;
  // Longest-first dispatch: a copy's solver work (rows x sweeps) in the previous step predicts this step's, and
  // workgroups are dispatched in index order, so handing the heavy copies to the lowest workgroup ids keeps a straggler
  // from starting last.  Each wave files its copy under a work bucket for the next launch -- one count and one bit set
  // per bucket, lpt_*_out, updated by atomics that return nothing -- and picks its copy from the previous launch's
  // buckets, heaviest bucket first (lpt_*_in; null = identity order): workgroup w steps the copy of the w-th set bit.
  // Which workgroup steps a copy has no effect on the copy's result.
  const int* lpt_count_in;          // [LPT_BUCKETS]
  const unsigned* lpt_mask_in;      // [LPT_BUCKETS][lpt_words]: bit e of a bucket's row = copy e is in the bucket
  int* lpt_count_out;
  unsigned* lpt_mask_out;
  int* lpt_count_clear;             // the tables the NEXT launch files into: zeroed by this one (counts by workgroup 0,
  unsigned* lpt_mask_clear;         // word i of every bucket's row by workgroup i)
  int lpt_words;                    // (n_env + 31) / 32
  // Sticky count of physics frames that ran into a cap (what MuJoCo reports as mjWARN_CONTACTFULL / mjWARN_CNSTRFULL):
  // [0] frames whose contact list was cut at nconmax, [1] frames whose row list was cut at njmax.  Not touched by
  // forward-only launches.  [2] workgroups that found no copy in the dispatch tables although the counts covered them
  // (cannot happen with consistent tables; mjrl_cap_overflows turns a non-zero count into an error).
  unsigned long long* overflow;
  // diagnostic: [n_env][3] per workgroup, in dispatch order: start and end of the wave on the constant 100 MHz clock
  // and the copy it stepped; null outside tools/timeline_probe.py
  unsigned long long* timeline;
  // diagnostic: k > 0 ends every wave right after stage k - 1 (ST_* order) without writing anything back, so that
  // hardware counters of launches cut at successive stages give the instruction mix of each stage (tools/stage_mix.py)
  int stop_after;
};
enum { LPT_BUCKETS = 16 };

__host__ __device__ inline int frame_doubles(const DevModel& m) { return 7 * m.nbody + 7 * m.ngeom + 1 + 2 * m.nconmax; }

// fused plugin ops: prog_i = {kind, i1..i7}, prog_f = {f0..f3}
enum {
  OP_LANGUAGE = 1,        // i1 action slot, i2 store slot, i3 extra-obs index: store = int(action); obs = other agent's store (0 if absent)
  OP_DIST_REWARD = 2,     // i1 target kind (0 body xipos, 1 geom xpos), i2 target id, i3 store slot (-1 none), i4 mode; f0 scale
                          //   mode 0: reward += scale * (-dist);  mode 1: reward += scale * (previous dist - dist), store = dist
  OP_DIST_DONE = 3,       // i1 target kind, i2 target id; f0 threshold: terminated |= dist < threshold
                          //   (both DIST ops) target kind 2: the agent's CURRENT TARGET -- i2 = tag, i5 = store slot that holds
                          //   its index in the tag's list (kept by an OP_TARGET); the op does nothing while that slot is empty
  OP_TARGET = 4,          // Target-seeking dynamics of Testing/EnvironmentDynamic.py:17-32 and Testing/Pick_Up_Dynamic.py:15-41:
                          //   i1 tag, i2 store slot of the current target's index, i3 store slot of the inventory (-1: none),
                          //   i4 extra-obs index, i5 store slot of the distance (-1: none); f0 threshold, f1 reward, f2 seed.
                          //   empty slot -> choose a target (and inventory = 0); dist(agent, target) < threshold -> toggle the
                          //   inventory, reward += f1, choose again, store the distance to the new target;
                          //   obs = xipos / xpos of the current target (3) [+ inventory]
  MAX_AGENT = 8
};

// The counter-based generator behind every on-device random choice (splitmix64 finaliser over a linear key): a pure
// function of (seed, global copy id, agent, episode << 32 | episode step, salt), so the host plugin of the same name
// (dynamics.py) and any shard of the batch draw the same numbers -- and a copy's episodes differ from one another (the
// reference draws with random.randint, Testing/EnvironmentDynamic.py:26-29, Pick_Up_Dynamic.py:28,38).
__host__ __device__ inline unsigned long long mix64(unsigned long long seed, unsigned long long env, unsigned long long agent,
                                                    unsigned long long step, unsigned long long salt) {
  unsigned long long z = seed * 0x9E3779B97F4A7C15ull + env * 0xBF58476D1CE4E5B9ull + agent * 0x94D049BB133111EBull +
                         step * 0xD6E8FEB86659FD93ull + salt * 0xA0761D6478BD642Full;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
__host__ __device__ inline int pick_of(unsigned long long z, int n) { return n > 0 ? (int)((z >> 33) % (unsigned long long)n) : 0; }

enum { ST_LOAD = 0, ST_KIN, ST_COM, ST_CRB, ST_FACTOR, ST_GEOM, ST_COLLIDE, ST_VEL, ST_SMOOTH, ST_ROWS, ST_PROJECT, ST_PGS,
       ST_SENSORS, ST_EULER, ST_STORE, ST_PGS_WARM, ST_PGS_LISTS, ST_PGS_SWEEPS, ST_ROWS_LIMITS, ST_ROWS_ADDR, ST_PGS_SETUP,
       ST_TAIL, N_STAMPS };

// The lane's own records of the model, fetched once per launch
struct LaneK {
  // lane as body
  int b_parent, b_depth, b_tree, b_dofadr, b_dofnum, b_jntadr, b_jntnum, b_subnum;
  int b_jnttype;      // type of the body's first joint (-1: none)
  int b_qposadr;      // its qpos address
  real b_q0;          // and its reference position qpos0 (hinge / slide)
  real b_mass;
  // lane as dof
  int d_parent, d_Madr, d_depth, d_body, d_descadr, d_descnum, d_act;
  real d_damping, d_armature;
  real d_stiffness, d_springref;      // joint spring of the lane's dof (hinge / slide; 0: none)
  int d_qposadr;                      // the coordinate it acts on
};

__host__ __device__ inline void load_lane_constants(const DevModel& m, int L, LaneK& k) {
  bool isb = L < m.nbody, isd = L < m.nv;
  int b = isb ? L : 0, d = isd ? L : 0;
  // (kinematic parent and depth: bodies welded to a jointless parent hang off the nearest ancestor that moves, mjcf.py)
  k.b_parent = m.body_kparent[b]; k.b_depth = isb ? m.body_kdepth[b] : -1; k.b_tree = m.body_treeid[b];
  k.b_dofadr = m.body_dofadr[b]; k.b_dofnum = isb ? m.body_dofnum[b] : 0; k.b_jntadr = m.body_jntadr[b];
  k.b_jntnum = isb ? m.body_jntnum[b] : 0; k.b_subnum = m.body_subtreenum[b]; k.b_mass = m.body_mass[b];
  k.b_jnttype = k.b_jntnum > 0 ? m.jnt_type[k.b_jntadr] : -1;
  k.b_qposadr = k.b_jntnum > 0 ? m.jnt_qposadr[k.b_jntadr] : 0;
  k.b_q0 = k.b_jntnum > 0 ? m.qpos0[k.b_qposadr] : 0.0;
  k.d_parent = m.dof_parentid[d]; k.d_Madr = m.dof_Madr[d]; k.d_depth = isd ? m.dof_depth[d] : -1;
  k.d_body = m.dof_bodyid[d]; k.d_descadr = m.dof_descadr[d]; k.d_descnum = isd ? m.dof_descnum[d] : 0;
  k.d_act = isd ? m.dof_actid[d] : -1; k.d_damping = m.dof_damping[d]; k.d_armature = m.dof_armature[d];
  k.d_stiffness = m.dof_stiffness[d]; k.d_springref = m.dof_springref[d]; k.d_qposadr = m.dof_qposadr[d];
}

// Structure tables kept in LDS as 16-bit words (order fixed by mjcf._kernel_schedules): the row build walks them
// with data-dependent indices, which from HBM would cost one exposed load latency per hop.
struct Tab {
  const unsigned char* t8;       // one byte per entry when every value fits (nM <= 256), else 16-bit words
  bool bytes;
  int Mcol, Madr, depth, dtree, lastdof, btree, gbody, gtype, gcondim;
  __device__ __forceinline__ int at(int i) const { return bytes ? (int)t8[i] : (int)((const unsigned short*)t8)[i]; }
  __device__ __forceinline__ int colid(int e) const { return at(Mcol + e); }
  __device__ __forceinline__ int madr(int d) const { return at(Madr + d); }
  __device__ __forceinline__ int ddepth(int d) const { return at(depth + d); }
  __device__ __forceinline__ int dof_tree(int d) const { return at(dtree + d); }
  __device__ __forceinline__ int body_lastdof(int b) const { return at(lastdof + b) - 1; }
  __device__ __forceinline__ int body_tree(int b) const { return at(btree + b) - 1; }
  __device__ __forceinline__ int geom_body(int g) const { return at(gbody + g); }
  __device__ __forceinline__ int geom_type(int g) const { return at(gtype + g); }
  __device__ __forceinline__ int geom_condim(int g) const { return at(gcondim + g); }
};

__device__ inline Tab make_tab(const DevModel& m, const Lay& l, const real* S) {
  Tab T;
  T.t8 = (const unsigned char*)(S + l.tab);
  T.bytes = tab_in_bytes(m);
  T.Mcol = 0; T.Madr = m.nM; T.depth = T.Madr + m.nv; T.dtree = T.depth + m.nv; T.lastdof = T.dtree + m.nv;
  T.btree = T.lastdof + m.nbody; T.gbody = T.btree + m.nbody; T.gtype = T.gbody + m.ngeom; T.gcondim = T.gtype + m.ngeom;
  return T;
}

// copy the launch-invariant tables into LDS (one coalesced sweep per launch).  The first TAB_EARLY x 64 entries are
// fetched into registers by tab_issue() -- at the top of the kernel, so that their round trip runs under the lookup of
// the copy's id -- and written by stage_constants(); a longer table takes a loop of batched loads for the rest.
enum { TAB_EARLY = 8 };
struct TabRegs { int v[TAB_EARLY]; };
__host__ __device__ __forceinline__ void tab_issue(const DevModel& m, int L, TabRegs& r) {
#pragma unroll
  for (int u = 0; u < TAB_EARLY; u++) {
    const int i = 64 * u + L;
    r.v[u] = 64 * u < m.ntab ? m.lds_tab[i < m.ntab ? i : 0] : 0;
  }
}
template <typename T>
__device__ __forceinline__ void tab_commit(const DevModel& m, T* t, int L, const TabRegs& r) {
#pragma unroll
  for (int u = 0; u < TAB_EARLY; u++) {
    const int i = 64 * u + L;
    if (i < m.ntab) t[i] = (T)r.v[u];
  }
  for (int base = 64 * TAB_EARLY; base < m.ntab; base += 64 * TAB_EARLY) {
    int v[TAB_EARLY];
#pragma unroll
    for (int u = 0; u < TAB_EARLY; u++) {
      const int i = base + 64 * u + L;
      v[u] = m.lds_tab[i < m.ntab ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < TAB_EARLY; u++) {
      const int i = base + 64 * u + L;
      if (i < m.ntab) t[i] = (T)v[u];
    }
  }
}
__device__ inline void stage_constants(const DevModel& m, const Lay& l, real* S, int L, const TabRegs& r) {
  if (tab_in_bytes(m)) tab_commit(m, (unsigned char*)(S + l.tab), L, r);
  else tab_commit(m, (unsigned short*)(S + l.tab), L, r);
  if (L == 0) S[l.zero] = 0.0;
}

__device__ __forceinline__ float int_as_float(int v) {
  float f;
  __builtin_memcpy(&f, &v, sizeof(f));
  return f;
}

// The lane's records in the tree-row lane map (tree t owns lanes 16t..16t+15, one dof per lane): the lane's dof and
// the factor entries it multiplies in each step of the register-resident triangular solves.
struct RowK {
  int dof;
  int depth;                    // of the lane's dof
  unsigned long long below;     // the dof itself and every dof below it, one bit per dof
  int eb[16], ef[16];
};

// slot of the lane's dof in a compact constraint row with chain code `chain`, -1 if the row does not touch the dof
__device__ __forceinline__ int row_slot(int chain, unsigned long long below, int depth) {
  int xp = chain & 63, dp = (chain >> 6) & 7, xq1 = (chain >> 9) & 127, dq = (chain >> 16) & 7;
  bool in_p = (below >> xp) & 1ull;
  bool in_q = xq1 != 0 && ((below >> ((xq1 - 1) & 63)) & 1ull);
  return in_p ? dp - depth : (in_q ? MAX_DOF_DEPTH + dq - depth : -1);
}

__host__ __device__ inline void load_row_constants(const DevModel& m, const Lay& l, int L, RowK& r) {
  r.dof = m.rowmap ? m.row_dof[L] : (L < m.nv ? L : -1);
  int d = r.dof >= 0 ? r.dof : 0;
  r.depth = m.dof_depth[d];
  r.below = r.dof >= 0 ? ((unsigned long long)(unsigned)m.dof_descmask[2 * d + 1] << 32) | (unsigned)m.dof_descmask[2 * d] : 0ull;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    // absolute LDS offsets of the factor entries (the factor always sits at l.LD); no entry -> the zero slot, so that
    // the solves need no predication
    int eb = m.rowmap ? m.solve_b[k * 64 + L] : -1, ef = m.rowmap ? m.solve_f[k * 64 + L] : -1;
    r.eb[k] = eb >= 0 ? l.LD + eb : l.zero;
    r.ef[k] = ef >= 0 ? l.LD + ef : l.zero;
  }
}

// ------------------------------------------------------------------ position stage
// The model constants of the first two stages (the lane's body: offset from its kinematic parent, first joint, inertial
// frame).  They are used once, right at the start of the step, so they are fetched with the prologue's other loads
// instead of costing each stage a round trip to L2 of its own.
struct KinK {
  V3 bpos; Quat bquat;         // offset from the kinematic parent
  V3 jaxis0, jpos0;            // the body's first joint
  V3 ipos; Quat iquat;         // inertial frame, principal inertias (centre-of-mass stage)
  real inertia[3];
};
__host__ __device__ __forceinline__ void load_kin_constants(const DevModel& m, int L, const LaneK& K, KinK& k) {
  const int b = L < m.nbody ? L : 0;
  k.bpos = ld3(m.body_kpos + 3 * b);
  k.bquat = ldq(m.body_kquat + 4 * b);
  const int j0 = K.b_jntnum > 0 ? K.b_jntadr : 0;       // (lanes without a joint read joint 0 and ignore it)
  k.jaxis0 = ld3(m.jnt_axis + 3 * j0);
  k.jpos0 = ld3(m.jnt_pos + 3 * j0);
  k.ipos = ld3(m.body_ipos + 3 * b);
  k.iquat = ldq(m.body_iquat + 4 * b);
  for (int i = 0; i < 3; i++) k.inertia[i] = m.body_inertia[3 * b + i];
}

__device__ inline void stage_kinematics(const DevModel& m, const Lay& l, const LaneK& K, const KinK& KK, real* S, int L) {
  if (L == 0) {
    st3(S + l.xpos, v3(0, 0, 0));
    Quat q; q.w = 1; q.x = q.y = q.z = 0;
    stq(S + l.xquat, q);
  }
  // Everything that does not depend on the parent's frame is done before the level loop, for all bodies at once: the
  // body's offset, and for its first joint (the only one of most bodies) the constants and the joint's own motion --
  // the rotation quaternion of a hinge (one sincos for the whole wave instead of one per tree level), the normalised
  // quaternion of a free joint.
  const V3 bpos = KK.bpos;
  const Quat bquat = KK.bquat;
  const int j0 = K.b_jntadr;
  const bool hasj = K.b_jntnum > 0;
  const int jt0 = K.b_jnttype, qa0 = K.b_qposadr;
  V3 jaxis0 = hasj ? KK.jaxis0 : v3(0, 0, 1), jpos0 = hasj ? KK.jpos0 : v3(0, 0, 0);
  real q0 = 0;
  V3 fpos = v3(0, 0, 0);
  Quat jq; jq.w = 1; jq.x = jq.y = jq.z = 0;
  if (jt0 == JNT_FREE) {
    fpos = ld3(S + l.qpos + qa0);
    jq = qnormalized(ldq(S + l.qpos + qa0 + 3));
  } else if (hasj) {
    q0 = S[l.qpos + qa0] - K.b_q0;
    if (jt0 == JNT_HINGE) jq = axis_angle(jaxis0, q0);
  }
  wv::sync();
  for (int lev = 1; lev <= m.maxkdepth; lev++) {
    if (K.b_depth == lev) {
      int b = L, p = K.b_parent;
      Quat pq = ldq(S + l.xquat + 4 * p);
      V3 pos = ld3(S + l.xpos + 3 * p) + rot(pq, bpos);
      Quat quat = qmul(pq, bquat);
      if (hasj) {
        if (jt0 == JNT_FREE) {
          pos = fpos;
          quat = jq;
          st3(S + l.xanchor + 3 * j0, pos);
          st3(S + l.xaxis + 3 * j0, rot(quat, jaxis0));
        } else {
          V3 anchor = pos + rot(quat, jpos0);
          V3 axis = rot(quat, jaxis0);
          st3(S + l.xanchor + 3 * j0, anchor);
          st3(S + l.xaxis + 3 * j0, axis);
          if (jt0 == JNT_HINGE) {
            quat = qmul(quat, jq);
            pos = anchor - rot(quat, jpos0);
          } else {
            pos = pos + axis * q0;
          }
        }
      }
      for (int k = 1; k < K.b_jntnum; k++) {           // further joints of the same body (rare)
        int j = K.b_jntadr + k, qa = m.jnt_qposadr[j];
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

// (lane as one coordinate of a tree's centre of mass, lane as joint: fetched while the kinematics run)
struct ComK { int root, n; real mass; int j_body, j_dofadr, j_type; };
__host__ __device__ __forceinline__ void load_com_constants(const DevModel& m, int L, ComK& c) {
  const int t = L < 3 * m.ntree ? L / 3 : 0;
  c.root = m.tree_rootbody[t];
  c.n = m.body_subtreenum[c.root];
  c.mass = m.body_subtreemass[c.root];
  const int j = L < m.njnt ? L : 0;
  c.j_body = m.njnt > 0 ? m.jnt_bodyid[j] : 0;
  c.j_dofadr = m.njnt > 0 ? m.jnt_dofadr[j] : 0;
  c.j_type = m.njnt > 0 ? m.jnt_type[j] : -1;
}

__device__ inline void stage_com_inertia(const DevModel& m, const Lay& l, const LaneK& K, const KinK& KK, const ComK& CK,
                                         real* S, int L) {
  const Tab T = make_tab(m, l, S);
  // mass-weighted centres of mass; a tree's bodies are the id range of its root (depth-first numbering)
  V3 xi = v3(0, 0, 0);
  Quat q; q.w = 1; q.x = q.y = q.z = 0;
  if (L > 0 && L < m.nbody) {
    q = ldq(S + l.xquat + 4 * L);
    xi = ld3(S + l.xpos + 3 * L) + rot(q, KK.ipos);
    st3(S + l.cinert + 10 * L, xi * K.b_mass);     // (scratch until the sums are taken; cinert is written after)
  }
  wv::sync();
  if (L < 3 * m.ntree) {
    int t = L / 3, k = L % 3, root = CK.root, n = CK.n;
    real acc = 0;
    for (int c = root; c < root + n; c++) acc += S[l.cinert + 10 * c + k];
    real mass = CK.mass;
    S[l.com + 3 * t + k] = mass < MJ_MINVAL ? S[l.xpos + 3 * root + k] : acc / mass;
  }
  if (L < 3) S[l.com + 3 * m.ntree + L] = 0;
  wv::sync();
  // body inertias about their tree's centre of mass
  if (L < m.nbody) {
    real ci[10];
    if (L == 0 || K.b_tree < 0) {
      for (int k = 0; k < 10; k++) ci[k] = 0;
    } else {
      M3 ximat = qmat(qmul(q, KK.iquat));
      inert_com(ci, KK.inertia, ximat, xi - ld3(S + l.com + 3 * K.b_tree), K.b_mass);
    }
    for (int k = 0; k < 10; k++) S[l.cinert + 10 * L + k] = ci[k];
  }
  // joint motion axes in the same frame
  if (L < m.njnt) {
    int j = L, b = CK.j_body, da = CK.j_dofadr;
    V3 off = ld3(S + l.com + 3 * T.body_tree(b)) - ld3(S + l.xanchor + 3 * j);     // (tree id from the LDS table)
    if (CK.j_type == JNT_FREE) {
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
      if (CK.j_type == JNT_HINGE) { st3(c, axis); st3(c + 3, cross(axis, off)); }
      else { st3(c, v3(0, 0, 0)); st3(c + 3, axis); }
    }
  }
  wv::sync();
}

// The unfactorised inertia matrix between the CRB stage and the integrator (M + h*diag(damping) is factorised there, a
// whole step of work later): element 64 k + L of the dof_Madr layout in lane L.  nM <= nv * MAX_DOF_DEPTH <= 512, i.e.
// at most 8 doubles per lane -- 3 for the 2-agent level, where the matrix used to make a round trip through a per-copy
// row in HBM (2.6 KB of traffic per copy and step, half of what the kernel moved).
enum { M_KEEP = 8 };
struct MKeep { real v[M_KEEP]; };
// (both inlined by force: through a call the eight values would be an object on the stack)
__device__ __forceinline__ void mkeep_take(const DevModel& m, const Lay& l, const real* S, int L, MKeep& Mk) {
#pragma unroll
  for (int k = 0; k < M_KEEP; k++) Mk.v[k] = (64 * k < m.nM) ? S[l.LD + (64 * k + L < m.nM ? 64 * k + L : 0)] : 0.0;
  wv::sync();          // (the factorisation that follows works in place)
}
__device__ __forceinline__ void mkeep_put(const DevModel& m, const Lay& l, real* S, int L, const MKeep& Mk) {
#pragma unroll
  for (int k = 0; k < M_KEEP; k++)
    if (64 * k < m.nM && 64 * k + L < m.nM) S[l.LD + 64 * k + L] = Mk.v[k];
}
__device__ inline void stage_crb(const DevModel& m, const Lay& l, const LaneK& K, real* S, int L) {
  const Tab T = make_tab(m, l, S);
  // composite inertia of every body that carries dofs: sum of cinert over its subtree (an id range)
  if (L > 0 && L < m.nbody && K.b_dofnum > 0) {
    real acc[10];
    for (int k = 0; k < 10; k++) acc[k] = S[l.cinert + 10 * L + k];
    for (int c = L + 1; c < L + K.b_subnum; c++)
      for (int k = 0; k < 10; k++) acc[k] += S[l.cinert + 10 * c + k];
    for (int k = 0; k < 10; k++) S[l.crb + 10 * L + k] = acc[k];
  }
  wv::sync();
  if (L < m.nv) {
    real buf[6], c[6], crb[10];
    for (int k = 0; k < 10; k++) crb[k] = S[l.crb + 10 * K.d_body + k];
    for (int k = 0; k < 6; k++) c[k] = S[l.cdof + 6 * L + k];
    inert_mul(buf, crb, c);
    for (int t = 0; t <= K.d_depth; t++) {
      int j = T.colid(K.d_Madr + t);          // (LDS structure table: a load from HBM per ancestor would sit on the chain)
      real cj[6];
      for (int k = 0; k < 6; k++) cj[k] = S[l.cdof + 6 * j + k];
      real v = dot6(cj, buf);
      if (t == 0) v = K.d_armature + v;
      S[l.LD + K.d_Madr + t] = v;
    }
  }
  wv::sync();
}

// In-place sparse L'DL of the matrix at S[ld..] (dof_Madr layout), Featherstone order inside every kinematic tree.
// Trees are independent, so up to two of them are eliminated side by side (32 lanes each); inside a tree the
// lanes cover the (ancestor a, offset t) pairs of the pivot row.  dinv gets 1/D.
// The schedule words of the first FACTOR_AHEAD elimination steps of a factorisation, fetched ahead of it (before the
// stage in front) so that the factorisation starts without a round trip to L2 of its own.
enum { FACTOR_AHEAD = 4 };
// (q / adr0 / num: the lane's tree in pass 0; q1 / adr1 / num1: in pass 1, when the model has more than two trees)
struct FactorRing { unsigned q[FACTOR_AHEAD], q1[FACTOR_AHEAD]; int adr0, num, adr1, num1; };
__host__ __device__ __forceinline__ void factor_ring_load(const DevModel& m, int L, int ps, unsigned* q, int& adr0, int& num) {
  const int groups = m.ntree > 1 ? 2 : 1, tree = ps * groups + L / (64 / groups);
  adr0 = m.tree_dofadr[tree < m.ntree ? tree : 0];
  num = tree < m.ntree ? m.tree_dofnum[tree] : 0;
  const int32_t* sched = m.factor_sched + (size_t)ps * m.maxtreedof * 64 + L;
#pragma unroll
  for (int u = 0; u < FACTOR_AHEAD; u++) {
    const int kk = m.maxtreedof - 1 - u;
    q[u] = kk >= 0 ? (unsigned)sched[kk * 64] : 0u;
  }
}
__host__ __device__ __forceinline__ void factor_prefetch(const DevModel& m, int L, FactorRing& r) {
  factor_ring_load(m, L, 0, r.q, r.adr0, r.num);
  if (m.npass > 1) factor_ring_load(m, L, 1, r.q1, r.adr1, r.num1);
  else {
#pragma unroll
    for (int u = 0; u < FACTOR_AHEAD; u++) r.q1[u] = 0u;
    r.adr1 = 0; r.num1 = 0;
  }
}

// One elimination step of one pass, in two halves around the step's fence: what it reads and computes, what it writes.
struct FactorStep { bool valid, live; int ijt, ki, t, slot; real val, tmp, rkk; };
__device__ __forceinline__ void factor_step_read(const real* S, int ld, int dinv, unsigned w, int kk, int adr0, int num,
                                                 FactorStep& f) {
  f.live = kk < num; f.valid = (w >> 26) & 1u;
  const int kkadr = w & 1023, a = (w >> 20) & 7;
  f.ijt = (w >> 10) & 1023; f.t = (w >> 23) & 7;
  f.ki = kkadr + 1 + a;
  f.slot = dinv + adr0 + kk;
  // the pivot D_kk is final when its step starts (only deeper dofs update it), so its reciprocal is taken from the
  // same read as the row scaling and the step needs no second LDS round trip.  The elimination multiplier is formed
  // with that reciprocal -- one division per step (a dozen instructions, four of them at quarter rate) instead of
  // MuJoCo's two; the oracle does the same.  (The schedule word carries the pivot's address in every lane of the
  // tree, also in lanes without a pair.)
  // (no branch on `valid` here: a lane without a pair computes on the entries its zeroed fields address and writes
  // nothing -- straight-line code lets the two passes' reads and divisions be scheduled into each other)
  const real dkk = S[ld + kkadr], ski = S[ld + f.ki], sij = S[ld + f.ijt], skt = S[ld + f.ki + f.t];
  f.rkk = 1.0 / dkk;
  f.tmp = ski * f.rkk;
  f.val = sij - f.tmp * skt;
}
__device__ __forceinline__ void factor_step_write(real* S, int ld, bool first_lane, const FactorStep& f) {
  if (f.valid) {
    S[ld + f.ijt] = f.val;
    if (f.t == 0) S[ld + f.ki] = f.tmp;
  }
  if (f.live && first_lane) S[f.slot] = f.rkk;
}

// Passes (two trees each) touch disjoint trees, so they are eliminated two passes at a time: step kk of both between the
// same two fences -- the second pass's LDS round trips and its division run under the first's (a model with four trees
// otherwise waits out 28 steps one after the other with a single wave on its SIMD).
__device__ inline void factor_ld(const DevModel& m, real* S, int ld, int dinv, int L, FactorRing ring) {
  const int groups = m.ntree > 1 ? 2 : 1, width = 64 / groups;
  const bool first_lane = L % width == 0;
  for (int ps = 0; ps < m.npass; ps += 2) {
    const bool two = ps + 1 < m.npass;
    if (ps > 0) {
      factor_ring_load(m, L, ps, ring.q, ring.adr0, ring.num);
      if (two) factor_ring_load(m, L, ps + 1, ring.q1, ring.adr1, ring.num1);
    }
    // the lane's (ancestor, offset) pair of every elimination step comes from the host-built schedule, fetched
    // FACTOR_AHEAD steps before its use (a step is two short LDS phases, far less than a round trip to L2)
    const int32_t* sched0 = m.factor_sched + (size_t)ps * m.maxtreedof * 64 + L;
    const int32_t* sched1 = sched0 + (size_t)m.maxtreedof * 64;
    for (int kk = m.maxtreedof - 1; kk >= 0; kk--) {
      const unsigned w0 = ring.q[0], w1 = ring.q1[0];
#pragma unroll
      for (int u = 0; u + 1 < FACTOR_AHEAD; u++) { ring.q[u] = ring.q[u + 1]; ring.q1[u] = ring.q1[u + 1]; }
      ring.q[FACTOR_AHEAD - 1] = kk >= FACTOR_AHEAD ? (unsigned)sched0[(kk - FACTOR_AHEAD) * 64] : 0u;
      ring.q1[FACTOR_AHEAD - 1] = (two && kk >= FACTOR_AHEAD) ? (unsigned)sched1[(kk - FACTOR_AHEAD) * 64] : 0u;
      FactorStep f0, f1;
      factor_step_read(S, ld, dinv, w0, kk, ring.adr0, ring.num, f0);
      if (two) factor_step_read(S, ld, dinv, w1, kk, ring.adr1, ring.num1, f1);
      wv::sync();
      factor_step_write(S, ld, first_lane, f0);
      if (two) factor_step_write(S, ld, first_lane, f1);
      wv::sync();
    }
  }
}

// x <- (selected factors of) M^-1 x for a vector held ONE DOF PER LANE in the tree-row lane map, entirely in
// registers: step kk broadcasts x of the tree's kk-th dof inside the tree's row of 16 lanes (DPP), every lane
// subtracts its factor entry times that value.  The factor entries are fetched from LDS up front, so the dependent
// chain is DPP + multiply + subtract only.  Backward: descendants before ancestors; forward: ancestors first.
__device__ inline real solve_rows(const DevModel& m, const RowK& R, const real* S, int ld, int dinv, real x, bool backward,
                                  bool scale, bool forward) {
  real lb[16], lf[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    lb[k] = backward ? S[R.eb[k]] : 0.0;      // (absolute offsets, factor at l.LD == ld; see load_row_constants)
    lf[k] = forward ? S[R.ef[k]] : 0.0;
  }
  real di = (scale && R.dof >= 0) ? S[dinv + R.dof] : 1.0;
  if (backward) {
#define MJ_BSTEP(KK) if (KK < m.maxtreedof) { real xk = wv::bcast16<KK>(x); x = x - lb[KK] * xk; }
    MJ_BSTEP(15) MJ_BSTEP(14) MJ_BSTEP(13) MJ_BSTEP(12) MJ_BSTEP(11) MJ_BSTEP(10) MJ_BSTEP(9) MJ_BSTEP(8)
    MJ_BSTEP(7) MJ_BSTEP(6) MJ_BSTEP(5) MJ_BSTEP(4) MJ_BSTEP(3) MJ_BSTEP(2) MJ_BSTEP(1)
#undef MJ_BSTEP
  }
  if (scale) x = x * di;
  if (forward) {
#define MJ_FSTEP(KK) if (KK < m.maxtreedof) { real xk = wv::bcast16<KK>(x); x = x - lf[KK] * xk; }
    MJ_FSTEP(0) MJ_FSTEP(1) MJ_FSTEP(2) MJ_FSTEP(3) MJ_FSTEP(4) MJ_FSTEP(5) MJ_FSTEP(6) MJ_FSTEP(7)
    MJ_FSTEP(8) MJ_FSTEP(9) MJ_FSTEP(10) MJ_FSTEP(11) MJ_FSTEP(12) MJ_FSTEP(13) MJ_FSTEP(14)
#undef MJ_FSTEP
  }
  return x;
}

// x <- (selected factors of) M^-1 x with the factor at ld/dinv; x lives in LDS at S[x..x+nv).  Level-parallel over
// dof depth; every dof applies its terms in the order the serial sweeps would (descendants descending, ancestors
// walking up).
__device__ inline void solve_ld(const DevModel& m, const LaneK& K, real* S, int ld, int dinv, int x, int L, bool backward,
                                bool scale, bool forward) {
  if (backward) {
    for (int lev = m.maxdofdepth - 1; lev >= 0; lev--) {
      if (K.d_depth == lev && K.d_descnum > 0) {
        real xi = S[x + L];
        for (int c = K.d_descnum - 1; c >= 0; c--)
          xi -= S[ld + m.desc_Madr[K.d_descadr + c]] * S[x + m.desc_row[K.d_descadr + c]];
        S[x + L] = xi;
      }
      wv::sync();
    }
  }
  if (scale) {
    if (L < m.nv) S[x + L] *= S[dinv + L];
    wv::sync();
  }
  if (forward) {
    for (int lev = 1; lev <= m.maxdofdepth; lev++) {
      if (K.d_depth == lev) {
        real xi = S[x + L];
        for (int t = 1; t <= K.d_depth; t++) xi -= S[ld + K.d_Madr + t] * S[x + m.M_colid[K.d_Madr + t]];
        S[x + L] = xi;
      }
      wv::sync();
    }
  }
}

// world frame of geom g from its body's frame.  The geom stage stores these for the collision stage only (scratch
// area of `u`); what needs a geom's frame after that -- rangefinders, the frame cache, a distance op -- calls this
// again and gets the same bits.
__device__ __forceinline__ void geom_frame(const DevModel& m, const Lay& l, const real* S, int g, V3& pos, Quat& quat) {
  int b = m.geom_bodyid[g];
  Quat bq = ldq(S + l.xquat + 4 * b);
  pos = ld3(S + l.xpos + 3 * b) + rot(bq, ld3(m.geom_pos + 3 * g));
  quat = qmul(bq, ldq(m.geom_quat + 4 * g));
}

// The ray caster's scene row of one copy (StepArgs::scene) from the body frames in LDS.
__host__ __device__ inline int scene_doubles(const DevModel& m) { return 12 * m.ngeom + 12 * m.ncam + 6 * m.nlight; }
__device__ inline void write_scene_row(const DevModel& m, const Lay& l, const real* S, int L, real* out) {
  for (int g = L; g < m.ngeom; g += 64) {
    V3 gp; Quat gq;
    geom_frame(m, l, S, g, gp, gq);
    st3(out + 3 * g, gp);
    M3 gm = qmat(gq);
    for (int k = 0; k < 9; k++) out[3 * m.ngeom + 9 * g + k] = gm.m[k];
  }
  for (int cam = L; cam < m.ncam; cam += 64) {
    const int body = m.cam_bodyid[cam];
    const Quat bq = ldq(S + l.xquat + 4 * body);
    st3(out + 12 * m.ngeom + 3 * cam, ld3(S + l.xpos + 3 * body) + rot(bq, ld3(m.cam_pos + 3 * cam)));
    M3 cm = qmat(qmul(bq, ldq(m.cam_quat + 4 * cam)));
    for (int k = 0; k < 9; k++) out[12 * m.ngeom + 3 * m.ncam + 9 * cam + k] = cm.m[k];
  }
  for (int li = L; li < m.nlight; li += 64) {        // the level's lights ride on their bodies
    const int body = m.light_bodyid[li];
    const Quat bq = ldq(S + l.xquat + 4 * body);
    st3(out + 12 * m.ngeom + 12 * m.ncam + 3 * li, ld3(S + l.xpos + 3 * body) + rot(bq, ld3(m.light_pos + 3 * li)));
    st3(out + 12 * m.ngeom + 12 * m.ncam + 3 * m.nlight + 3 * li, rot(bq, ld3(m.light_dir + 3 * li)));
  }
}

// The lane's geom record and the candidate pairs of the first broad-phase chunks, fetched two stages ahead (before the
// inertia matrix is built and factorised): the geom and collision stages then start on registers.
enum { PAIR_AHEAD = 3 };
struct GeomK {
  int body; V3 pos; Quat quat;
  real size[3];                           // geom_size[L], [L + 64], [L + 128] (a longer table takes a loop)
  int word[PAIR_AHEAD], reach[PAIR_AHEAD];
  int chunk_info;                         // lane c: kind and tree pair of chunk c of the pair list
  int tp_a, tp_b; real tp_reach;          // lane k: root bodies and reach of tree pair k
};
__host__ __device__ __forceinline__ void load_geom_constants(const DevModel& m, int L, GeomK& g) {
  const int i = L < m.ngeom ? L : 0;
  g.body = m.geom_bodyid[i];
  g.pos = ld3(m.geom_pos + 3 * i);
  g.quat = ldq(m.geom_quat + 4 * i);
#pragma unroll
  for (int u = 0; u < 3; u++) g.size[u] = 64 * u < 3 * m.ngeom ? m.geom_size[64 * u + L < 3 * m.ngeom ? 64 * u + L : 0] : 0.0;
#pragma unroll
  for (int u = 0; u < PAIR_AHEAD; u++) {
    const int p = 64 * u + L, q = p < m.npair ? p : 0;
    g.word[u] = 64 * u < m.npair ? m.pair_word[q] : 0;
    g.reach[u] = 64 * u < m.npair ? m.pair_reach[q] : 0;
  }
  g.chunk_info = m.nchunk > 0 ? m.chunk_info[L < m.nchunk ? L : 0] : 0;
  const int k = L < m.ntp ? L : 0;
  g.tp_a = m.ntp > 0 ? m.tp_root[2 * k] : 0;
  g.tp_b = m.ntp > 0 ? m.tp_root[2 * k + 1] : 0;
  g.tp_reach = m.ntp > 0 ? m.tp_reach[k] : 0.0;
}

__device__ inline void stage_geoms(const DevModel& m, const Lay& l, const GeomK& G, real* S, int L) {
  if (L < m.ngeom) {
    // (the arithmetic of geom_frame() on the prefetched record)
    Quat bq = ldq(S + l.xquat + 4 * G.body);
    V3 pos = ld3(S + l.xpos + 3 * G.body) + rot(bq, G.pos);
    Quat quat = qmul(bq, G.quat);
    st3(S + l.gpos + 3 * L, pos);
    stq(S + l.gquat + 4 * L, quat);
  }
  for (int g = 64 + L; g < m.ngeom; g += 64) {      // (geoms past the 64th: their records straight from the model)
    V3 pos; Quat quat;
    geom_frame(m, l, S, g, pos, quat);
    st3(S + l.gpos + 3 * g, pos);
    stq(S + l.gquat + 4 * g, quat);
  }
  // the geom sizes go next to the work items of the collision stage (scratch area of `u`, free since the composite
  // inertias were consumed)
#pragma unroll
  for (int u = 0; u < 3; u++)
    if (64 * u + L < 3 * m.ngeom) S[l.gsize + 64 * u + L] = G.size[u];
  for (int i = 192 + L; i < 3 * m.ngeom; i += 64) S[l.gsize + i] = m.geom_size[i];
  wv::sync();
}

// ------------------------------------------------------------------ collision
__device__ inline void stage_collision(const DevModel& m, const Lay& l, const GeomK& G, real* S, int L) {
  int* I = (int*)(S + l.ints);
  int nitem = 0, warn = 0;
  // Broad phase.  The candidate pairs come in chunks of ONE kind of test each (mjcf._pair_layout: plane pairs, (sphere |
  // capsule)-box pairs, box-box pairs, bounding-sphere pairs; padded with empty entries), so a chunk runs one test and a
  // scalar branch picks it -- a chunk of mixed kinds ran every kind's code under lane masks.  The bounding-sphere pairs
  // between two kinematic trees form a block per pair of trees, and a block is skipped when the trees themselves are out
  // of each other's reach (root body positions against a reach no joint configuration exceeds): its pairs would all fail.
  // Every surviving pair expands into its narrow-phase work items, in list order.
  // one packed word + one float per candidate pair; those of the next PAIR_AHEAD chunks are in flight while this chunk
  // is tested (a chunk without a survivor is a few dozen instructions, much less than a round trip to L2)
  const int info_reg = G.chunk_info;                // lane c: kind and tree pair of chunk c (first 64 chunks)
  unsigned long long tp_live = ~0ull;
  if (m.ntp > 0) {
    const V3 d = ld3(S + l.xpos + 3 * G.tp_a) - ld3(S + l.xpos + 3 * G.tp_b);
    tp_live = ~wv::ballot(L < m.ntp && dot(d, d) > G.tp_reach * G.tp_reach);
  }
  int wring[PAIR_AHEAD], rring[PAIR_AHEAD];
#pragma unroll
  for (int u = 0; u < PAIR_AHEAD; u++) { wring[u] = G.word[u]; rring[u] = G.reach[u]; }
  // the two geom centres of the lane's pair are read from LDS one chunk ahead as well (a wave alone on its SIMD -- the
  // 4-agent arena -- otherwise waits out an LDS round trip at the head of every chunk)
  V3 c1 = ld3(S + l.gpos + 3 * (wring[0] & 255)), c2 = ld3(S + l.gpos + 3 * ((wring[0] >> 8) & 255));
  // (the specialised kernel unrolls this loop completely: the prefetch ring then lives in fixed registers.  As a loop
  // the ring is shifted by register moves, a move has to wait for the load that fills its source, and every chunk began
  // with a wait for the loads issued one chunk earlier -- most of a chunk's time.)
#ifdef MJRL_SPEC
#pragma unroll
#endif
  for (int base = 0; base < m.npair; base += 64) {
    int p = base + L, items = 0;
    int word = wring[0];
    real bound = (real)int_as_float(rring[0]);
#pragma unroll
    for (int u = 0; u + 1 < PAIR_AHEAD; u++) { wring[u] = wring[u + 1]; rring[u] = rring[u + 1]; }
    {
      const int pn = p + 64 * PAIR_AHEAD, q = pn < m.npair ? pn : 0;
      const bool more = base + 64 * PAIR_AHEAD < m.npair;
      wring[PAIR_AHEAD - 1] = more ? m.pair_word[q] : 0;
      rring[PAIR_AHEAD - 1] = more ? m.pair_reach[q] : 0;
    }
    const V3 p1 = c1, p2 = c2;
    c1 = ld3(S + l.gpos + 3 * (wring[0] & 255)); c2 = ld3(S + l.gpos + 3 * ((wring[0] >> 8) & 255));   // (past the list / padding: geom 0, unused)
    const int chunk = base >> 6;
    const int info = chunk < 64 ? wv::lane_int(info_reg, chunk) : wv::first_int(m.chunk_info[chunk]);
    // (the kind follows from the chunk's place in the list -- sizes of the model, so constants in a specialised kernel,
    // whose unrolled chunks then hold one test each and no dispatch)
    const int kind = chunk < m.nchunk_plane ? 0 : (chunk < m.nchunk_plane + m.nchunk_box ? 1 :
                     (chunk < m.nchunk_plane + m.nchunk_box + m.nchunk_boxbox ? 2 : 3));
    const int tp = info >> 8;
    if (tp > 0 && tp <= 64 && !((tp_live >> (tp - 1)) & 1ull)) continue;      // the two trees are out of reach of each other
    const bool real_pair = word >= 0;                 // (bit 31: a padding entry)
    const int g1 = word & 255, g2 = (word >> 8) & 255;
    const V3 dif = p2 - p1;
    bool pass = false;
    // (bit 24 of the word: the frame this test needs is the identity rotation for good -- the arena's floor and walls --,
    // so its matrix is neither built nor applied; with an exact identity the general form gives the same bits)
    const bool fixed = (word >> 24) & 1;
    if (kind == 3) {                                  // bounding spheres (bound = rb1 + rb2 + margin)
      pass = real_pair && !(dot(dif, dif) > bound * bound);
    } else if (kind == 0) {                           // a plane: signed distance of the other geom's bounding sphere
      if (real_pair) pass = !((fixed ? dif.z : dot(dif, col(qmat(ldq(S + l.gquat + 4 * g1)), 2))) > bound);
    } else if (kind == 1) {
      // geom1's bounding sphere against the box itself: a long wall's bounding sphere would cover the whole arena
      if (real_pair) {
        V3 loc = fixed ? dif * -1.0 : mulT(qmat(ldq(S + l.gquat + 4 * g2)), dif * -1.0), bs = ld3(S + l.gsize + 3 * g2);
        real ex = fmax(fabs(loc.x) - bs.x, 0.0), ey = fmax(fabs(loc.y) - bs.y, 0.0), ez = fmax(fabs(loc.z) - bs.z, 0.0);
        pass = !(ex * ex + ey * ey + ez * ez > bound * bound);
      }
    } else if (m.pair_kmax >= 16) {                   // (compiled out of a specialised kernel whose level has no box-box pair)
      // box-box, the same both ways: each box's bounding sphere (+ margin) against the other box (bound = rb1 + rb2 + margin)
      pass = real_pair && !(dot(dif, dif) > bound * bound);
      if (pass) {
        const real rb1 = m.geom_rbound[g1], rb2 = m.geom_rbound[g2];
        V3 loc = mulT(qmat(ldq(S + l.gquat + 4 * g2)), dif * -1.0), bs = ld3(S + l.gsize + 3 * g2);
        real ex = fmax(fabs(loc.x) - bs.x, 0.0), ey = fmax(fabs(loc.y) - bs.y, 0.0), ez = fmax(fabs(loc.z) - bs.z, 0.0);
        const real r1 = bound - rb2;
        pass = !(ex * ex + ey * ey + ez * ez > r1 * r1);
        loc = mulT(qmat(ldq(S + l.gquat + 4 * g1)), dif); bs = ld3(S + l.gsize + 3 * g1);
        ex = fmax(fabs(loc.x) - bs.x, 0.0); ey = fmax(fabs(loc.y) - bs.y, 0.0); ez = fmax(fabs(loc.z) - bs.z, 0.0);
        const real r2 = bound - rb1;
        pass = pass && !(ex * ex + ey * ey + ez * ez > r2 * r2);
      }
    }
    if (pass) {
      items = 1 << ((word >> 25) & 7);                 // work items of the pair's types (pair_items), from the word
      if ((word >> 28) & 1) {
        // only (numerically) parallel capsules need the four end-point candidates
        real c = dot(col(qmat(ldq(S + l.gquat + 4 * g1)), 2), col(qmat(ldq(S + l.gquat + 4 * g2)), 2));
        if (fabs(1.0 - c * c) >= 1e-12) items = 1;
      }
    }
    // exclusive prefix of the item counts (1, 2, 4, 8 or 16 per lane) from ballots; most chunks have no survivor
    unsigned long long b1 = wv::ballot(items >= 1);
    if (b1 == 0ull) continue;
    unsigned long long lower = (1ull << L) - 1ull;
    unsigned long long b2 = wv::ballot(items >= 2), b4 = wv::ballot(items >= 4), b8 = wv::ballot(items >= 8);
    unsigned long long b16 = m.pair_kmax >= 16 ? wv::ballot(items >= 16) : 0ull;
    int off = nitem + wv::popc(b1 & lower) + wv::popc(b2 & lower) + 2 * wv::popc(b4 & lower) + 4 * wv::popc(b8 & lower) +
              8 * wv::popc(b16 & lower);
    for (int k = 0; k < items; k++)
      if (off + k < m.nitemmax) I[l.i_item + off + k] = (p << 4) | k;
    nitem += wv::popc(b1) + wv::popc(b2) + 2 * wv::popc(b4) + 4 * wv::popc(b8) + 8 * wv::popc(b16);
  }
  if (nitem > m.nitemmax) { nitem = m.nitemmax; warn |= 4; }
  wv::sync();
  // narrow phase: one lane per work item; contacts come out in (pair, item) order
  int ncon = 0;
  for (int base = 0; base < nitem; base += 64) {
    int it = base + L;
    bool hit = false;
    RawCon rc;
    int g1 = 0, g2 = 0;
    real margin = 0, gap = 0, mu = 0;
    if (it < nitem) {
      int code = I[l.i_item + it], p = code >> 4, k = code & 15;
      // (one round trip for everything a contact takes from the model: the pair's record holds the larger gap and the
      // larger sliding friction of its two geoms, fmax(geom_gap[g1], geom_gap[g2]) and fmax(geom_friction[3 g1],
      // geom_friction[3 g2]) evaluated once by the model compiler)
      int word = m.pair_word[p];
      margin = m.pair_margin[p];
      gap = m.pair_gap[p];
      mu = m.pair_mu[p];
      g1 = word & 255; g2 = (word >> 8) & 255;
      hit = collide_item((word >> 16) & 15, (word >> 20) & 15, ld3(S + l.gpos + 3 * g1), qmat(ldq(S + l.gquat + 4 * g1)),
                         ld3(S + l.gsize + 3 * g1), ld3(S + l.gpos + 3 * g2), qmat(ldq(S + l.gquat + 4 * g2)),
                         ld3(S + l.gsize + 3 * g2), margin, k, rc, m.pair_kmax >= 16);
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
        C[CON_MU] = mu;
        I[l.i_cong1 + slot] = g1;
        I[l.i_cong2 + slot] = g2;
      }
    }
    ncon += wv::popc(mask);
  }
  if (ncon > m.nconmax) { ncon = m.nconmax; warn |= 1; }
  if (L == 0) { I[I_NCON] = ncon; I[I_WARN] = warn; I[I_NITEM] = nitem; }
  wv::sync();
}

// ------------------------------------------------------------------ velocity stage (comVel + RNE bias)
__device__ inline void stage_velocity(const DevModel& m, const Lay& l, const LaneK& K, real* S, int L, bool with_acc) {
  // with_acc == false: cvel, cdof_dot, cacc (bias accelerations), body forces -> qfrc_bias
  // with_acc == true : cacc including cdof*qacc only (for the accelerometer), nothing else is touched
  if (L == 0) {
    for (int r = 0; r < 6; r++) { if (!with_acc) S[l.cvel + r] = 0; }
    S[l.cacc + 0] = S[l.cacc + 1] = S[l.cacc + 2] = 0;
    S[l.cacc + 3] = -m.gravity_x; S[l.cacc + 4] = -m.gravity_y; S[l.cacc + 5] = -m.gravity_z;
    if (!with_acc) for (int r = 0; r < 6; r++) S[l.crb + r] = 0;
  }
  wv::sync();
  real cvel[6] = {0, 0, 0, 0, 0, 0}, cacc[6] = {0, 0, 0, 0, 0, 0};     // the lane's body, kept for the force pass below
  for (int lev = 1; lev <= m.maxkdepth; lev++) {
    if (K.b_depth == lev) {
      int b = L, p = K.b_parent;
      for (int r = 0; r < 6; r++) { cvel[r] = S[l.cvel + 6 * p + r]; cacc[r] = S[l.cacc + 6 * p + r]; }
      int da = K.b_dofadr;
      if (!with_acc) {
        for (int k = 0; k < K.b_jntnum; k++) {
          int j = K.b_jntadr + k;
          if ((k == 0 ? K.b_jnttype : m.jnt_type[j]) == JNT_FREE) {
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
      da = K.b_dofadr;
      for (int t = 0; t < K.b_dofnum; t++)
        for (int r = 0; r < 6; r++) {
          cacc[r] += S[l.cdofdot + 6 * (da + t) + r] * S[l.qvel + da + t];
          if (with_acc) cacc[r] += S[l.cdof + 6 * (da + t) + r] * S[l.qacc + da + t];
        }
      for (int r = 0; r < 6; r++) S[l.cacc + 6 * b + r] = cacc[r];
    }
    wv::sync();
  }
  if (with_acc) return;
  // body force  I*a + v x* (I*v), every body at once from the velocities and accelerations its lane still holds
  // (crb's rows are free again: they hold the body forces)
  if (L > 0 && L < m.nbody) {
    real ci[10], t1[6], t2[6], t3[6];
    for (int k = 0; k < 10; k++) ci[k] = S[l.cinert + 10 * L + k];
    inert_mul(t1, ci, cacc);
    inert_mul(t2, ci, cvel);
    cross_force(t3, cvel, t2);
    for (int r = 0; r < 6; r++) S[l.crb + 10 * L + r] = t1[r] + t3[r];
  }
  wv::sync();
  // force transmitted by each dof-carrying body = sum of the body forces over its subtree; project on its dofs
  if (L > 0 && L < m.nbody && K.b_dofnum > 0) {
    real f[6];
    for (int r = 0; r < 6; r++) f[r] = S[l.crb + 10 * L + r];
    for (int c = L + 1; c < L + K.b_subnum; c++)
      for (int r = 0; r < 6; r++) f[r] += S[l.crb + 10 * c + r];
    for (int t = 0; t < K.b_dofnum; t++) {
      real c[6];
      for (int r = 0; r < 6; r++) c[r] = S[l.cdof + 6 * (K.b_dofadr + t) + r];
      S[l.bias + K.b_dofadr + t] = dot6(c, f);
    }
  }
  wv::sync();
}

// What the two stages after the velocity stage read from the model for the lane (as dof: its actuator; as joint-limit
// item: the joint's range), fetched before the velocity stage
struct ActK {
  int limited; real lo, hi, gear;                       // the lane's dof's actuator (K.d_act >= 0)
  int j_limited, j_type, j_qposadr; real j_range, j_margin;   // limit item L: joint L >> 1, lower side first
};
__host__ __device__ __forceinline__ void load_act_constants(const DevModel& m, int L, const LaneK& K, ActK& A) {
  const int u = K.d_act >= 0 ? K.d_act : 0;
  A.limited = m.nu > 0 ? m.act_ctrllimited[u] : 0;
  A.lo = m.nu > 0 ? m.act_ctrlrange[2 * u] : 0.0;
  A.hi = m.nu > 0 ? m.act_ctrlrange[2 * u + 1] : 0.0;
  A.gear = m.nu > 0 ? m.act_gear[u] : 0.0;
  const int it = L < 2 * m.njnt ? L : 0, j = it >> 1, side = (it & 1) ? 1 : -1;
  A.j_limited = m.njnt > 0 ? m.jnt_limited[j] : 0;
  A.j_type = m.njnt > 0 ? m.jnt_type[j] : -1;
  A.j_qposadr = m.njnt > 0 ? m.jnt_qposadr[j] : 0;
  A.j_range = m.njnt > 0 ? m.jnt_range[2 * j + (side + 1) / 2] : 0.0;
  A.j_margin = m.njnt > 0 ? m.jnt_margin[j] : 0.0;
}

// What the sensor stage reads from the model for the lane: as sensor (lane s: the sensor's record, its site's, the
// site's body) and as ray target (lane g: geom g's record for the rangefinders).
struct SensK {
  int site, adr, type, body;          // sensor s0 + L: its site, its address in sensordata, its type, the site's body
  real cut, size;                     // its cutoff, the site's size[0] (touch sensors)
  V3 pos; Quat quat;                  // the site's frame in its body
  int g_body, g_type;                 // geom L as a ray target
  real g_alpha, g_rb;
  V3 g_size, g_pos; Quat g_quat;
};
__host__ __device__ __forceinline__ void load_sensor_constants(const DevModel& m, int s0, int L, SensK& k) {
  const int sl = s0 + L < m.nsensor ? s0 + L : 0;
  k.site = m.nsensor > 0 ? m.sensor_objid[sl] : 0; k.adr = m.nsensor > 0 ? m.sensor_adr[sl] : 0;
  k.type = m.nsensor > 0 ? m.sensor_type[sl] : -1;
  k.cut = m.nsensor > 0 ? m.sensor_cutoff[sl] : 0.0;
  const int gl = L < m.ngeom ? L : 0;
  k.g_alpha = m.geom_rgba[4 * gl + 3];
  k.g_body = m.geom_bodyid[gl]; k.g_type = m.geom_type[gl];
  k.g_rb = m.geom_rbound[gl];
  k.g_size = ld3(m.geom_size + 3 * gl); k.g_pos = ld3(m.geom_pos + 3 * gl);
  k.g_quat = ldq(m.geom_quat + 4 * gl);
  k.body = m.nsite > 0 ? m.site_bodyid[k.site] : 0;
  k.pos = m.nsite > 0 ? ld3(m.site_pos + 3 * k.site) : v3(0, 0, 0);
  if (m.nsite > 0) k.quat = ldq(m.site_quat + 4 * k.site);
  else { k.quat.w = 1; k.quat.x = k.quat.y = k.quat.z = 0; }
  k.size = m.nsite > 0 ? m.site_size[3 * k.site] : 0.0;
}

// ------------------------------------------------------------------ lane records
// Everything above that a lane reads from the MODEL for itself -- its body's, dof's, joint's, geom's records, its rows of
// the solve / factor schedules, its slice of the structure tables -- depends on the lane id alone, never on the copy.
// Fetched field by field it was ninety loads per wave and launch with three to five vector instructions of address
// arithmetic each (a 64-bit base + constant + 4 x lane per table), a quarter of them behind another load (joint of the
// body -> type of that joint -> reference position of its coordinate): 740 vector instructions of the prologue, a tenth
// of them doing anything.  mjrl_create runs the same loader functions on the host, once, for lanes 0..63 and lays the
// results out as records: quad q of a group of lane L at int4[(first quad of the group + q) * 64 + L] -- a wave's fetch of
// one quad is 1 KB of consecutive bytes, the address one shift of the lane id, and one group costs ceil(bytes / 16)
// 16-byte loads.  Groups are fetched where their fields used to be (the prologue; ahead of the geom / factor / smooth /
// integrator stages).
struct EulerJ { int qa, da, jtype; };                  // lane as joint: what the integrator reads (load_euler_constants)
struct RecA { LaneK K; RowK RK; TabRegs TR; KinK KK; ComK CK; };
struct RecG { GeomK GK; };
struct RecR { FactorRing ring; };
struct RecC { ActK AK; };
struct RecE { EulerJ EJ; };
struct RecS { SensK SK; };                             // (sensors 0..63 and geoms 0..63; later chunks read the model)
template <typename T> constexpr int rec_quads() { return (int)((sizeof(T) + 15) / 16); }
enum { REC_A = 0, REC_G = REC_A + rec_quads<RecA>(), REC_R = REC_G + rec_quads<RecG>(), REC_C = REC_R + rec_quads<RecR>(),
       REC_E = REC_C + rec_quads<RecC>(), REC_S = REC_E + rec_quads<RecE>(), REC_QUADS = REC_S + rec_quads<RecS>() };
enum { LANE_REC_INTS = REC_QUADS * 64 * 4 };

// host: the table of a model (LANE_REC_INTS ints, zero-filled by the caller)
inline void build_lane_records(const DevModel& m, const Lay& l, int32_t* out) {
  auto put = [&](int first, const void* rec, size_t nbytes, int L) {
    // (memcpy, not a read through an int pointer: the records hold doubles, and under type-based alias analysis such a
    // read may be moved ahead of the loader's stores -- at -O3 the host compiler did, and the sensors' site size reached
    // the GPU as 0)
    for (size_t i = 0; i < nbytes / 4; i++)
      __builtin_memcpy(&out[((size_t)(first + i / 4) * 64 + L) * 4 + i % 4], (const char*)rec + 4 * i, 4);
  };
  for (int L = 0; L < 64; L++) {
    // (zero-filled first: padding bytes of the structs must not differ from run to run)
    RecA A; __builtin_memset(&A, 0, sizeof(A));
    load_lane_constants(m, L, A.K);
    load_row_constants(m, l, L, A.RK);
    tab_issue(m, L, A.TR);
    load_kin_constants(m, L, A.K, A.KK);
    load_com_constants(m, L, A.CK);
    put(REC_A, &A, sizeof(A), L);
    RecG G; __builtin_memset(&G, 0, sizeof(G));
    load_geom_constants(m, L, G.GK);
    put(REC_G, &G, sizeof(G), L);
    RecR R; __builtin_memset(&R, 0, sizeof(R));
    factor_prefetch(m, L, R.ring);
    put(REC_R, &R, sizeof(R), L);
    RecC C; __builtin_memset(&C, 0, sizeof(C));
    load_act_constants(m, L, A.K, C.AK);
    put(REC_C, &C, sizeof(C), L);
    RecE E; __builtin_memset(&E, 0, sizeof(E));
    const int j = L < m.njnt ? L : 0;
    E.EJ.qa = m.njnt > 0 ? m.jnt_qposadr[j] : 0;
    E.EJ.da = m.njnt > 0 ? m.jnt_dofadr[j] : 0;
    E.EJ.jtype = m.njnt > 0 ? m.jnt_type[j] : -1;
    put(REC_E, &E, sizeof(E), L);
    RecS SS; __builtin_memset(&SS, 0, sizeof(SS));
    load_sensor_constants(m, 0, L, SS.SK);
    put(REC_S, &SS, sizeof(SS), L);
  }
}

// device: one group of the lane's records into registers
typedef int rec_quad __attribute__((vector_size(16)));       // (the GNU spelling: the sanitized emulation is built by g++)
template <typename T>
__device__ __forceinline__ void fetch_lane_record(const int32_t* table, int first, int L, T& out) {
  static_assert(sizeof(T) % 4 == 0, "records are whole words");
  constexpr int N = rec_quads<T>();
  const rec_quad MJRL_GLOBAL* t = (const rec_quad MJRL_GLOBAL*)table + (size_t)first * 64 + L;
  rec_quad q[N];
#pragma unroll
  for (int k = 0; k < N; k++) q[k] = t[64 * k];
  __builtin_memcpy(&out, q, sizeof(T));
}

// qfrc_smooth = passive - bias + actuator ; qacc_smooth = M^-1 qfrc_smooth
__device__ inline void stage_smooth(const DevModel& m, const Lay& l, const LaneK& K, const RowK& R, const ActK& A, real* S,
                                    int L) {
  if (L < m.nv) {
    real act = 0;
    if (K.d_act >= 0) {
      int u = K.d_act;
      real c = S[l.ctrl + u];
      if (A.limited) c = fmin(fmax(c, A.lo), A.hi);
      act = A.gear * c;
    } else if (K.d_act == -2) {
      for (int u = 0; u < m.nu; u++) {
        if (m.act_dofid[u] != L) continue;
        real c = S[l.ctrl + u];
        if (m.act_ctrllimited[u]) c = fmin(fmax(c, m.act_ctrlrange[2 * u]), m.act_ctrlrange[2 * u + 1]);
        act += m.act_gear[u] * c;
      }
    }
    // (joint damping, and the joint's spring -stiffness (q - springref): mj_passive; the oracle's two statements)
    real passive = -K.d_damping * S[l.qvel + L];
    passive -= K.d_stiffness * (S[l.qpos + K.d_qposadr] - K.d_springref);
    real sm = passive - S[l.bias + L] + act;
    S[l.smooth + L] = sm;
    S[l.qaccs + L] = sm;
  }
  wv::sync();
  if (m.rowmap) {
    real x = solve_rows(m, R, S, l.LD, l.Dinv, R.dof >= 0 ? S[l.qaccs + R.dof] : 0.0, true, true, true);
    wv::sync();
    if (R.dof >= 0) S[l.qaccs + R.dof] = x;
    wv::sync();
  } else {
    solve_ld(m, K, S, l.LD, l.Dinv, l.qaccs, L, true, true, true);
  }
}

// ------------------------------------------------------------------ constraint rows
// One lane per row builds the Jacobian row, its reference acceleration and regularisation, and (unless a debug
// dump of the raw rows was requested) immediately projects it:  J <- J L^-1  by back substitution restricted to the
// row's own dof chains, and AR_ii = sum_d B_id^2 / D_d + R_i.  A row only touches the dof chains of its (at most
// two) bodies; the chains are read from the LDS structure tables.
template <bool DIAG>
__device__ inline void stage_rows(const DevModel& m, const Lay& l, const ActK& A, real* S, int L, bool project,
                                  Stamps* stamps) {
#define MJ_SUBSTAMP(k)                                                     \
  if constexpr (DIAG) if (stamps) {                                        \
    unsigned long long t_now = wv::clock();                                \
    if (L == (k)) stamps->mine += t_now - stamps->prev;                    \
    stamps->prev = t_now;                                                  \
  }
  int* I = (int*)(S + l.ints);
  const Tab T = make_tab(m, l, S);
  int ncon = I[I_NCON], warn = I[I_WARN];
  // joint limits: item = (joint, side), lower side first
  int nlim = 0;
  for (int base = 0; base < 2 * m.njnt; base += 64) {
    int it = base + L, j = it >> 1, side = (it & 1) ? 1 : -1;
    bool active = false;
    real dist = 0;
    // (the first 64 items test the prefetched record of their lane)
    const int limited = base == 0 ? A.j_limited : (it < 2 * m.njnt ? m.jnt_limited[j] : 0);
    const int jtype = base == 0 ? A.j_type : (it < 2 * m.njnt ? m.jnt_type[j] : -1);
    if (it < 2 * m.njnt && limited && (jtype == JNT_HINGE || jtype == JNT_SLIDE)) {
      real value = S[l.qpos + (base == 0 ? A.j_qposadr : m.jnt_qposadr[j])];
      dist = side * ((base == 0 ? A.j_range : m.jnt_range[2 * j + (side + 1) / 2]) - value);
      active = dist < (base == 0 ? A.j_margin : m.jnt_margin[j]);
    }
    unsigned long long mask = wv::ballot(active);
    if (active) {
      int r = nlim + wv::popc(mask & ((1ull << L) - 1ull));
      if (r < m.njmax) {
        I[l.i_rowid + r] = -(it + 1);          // negative: limit row, item id it
        S[l.row + ROW_STRIDE * r + ROW_F] = dist;
      }
    }
    nlim += wv::popc(mask);
  }
  if (nlim > m.njmax) { nlim = m.njmax; warn |= 2; }
  MJ_SUBSTAMP(ST_ROWS_LIMITS)
  // contacts: one lane per contact, pyramid addresses by prefix sums of the row counts (1, 2 or 4 rows each)
  {
    int rows = 0;
    if (L < ncon) {
      const real* C = S + l.con + CON_STRIDE * L;
      int g1 = I[l.i_cong1 + L], g2 = I[l.i_cong2 + L];
      int dim = T.geom_condim(g1) > T.geom_condim(g2) ? T.geom_condim(g1) : T.geom_condim(g2);
      if (C[CON_DIST] < C[CON_INCL]) rows = dim == 1 ? 1 : 2 * ((dim < 3 ? dim : 3) - 1);
    }
    unsigned long long lower = (1ull << L) - 1ull;
    unsigned long long b1 = wv::ballot(rows >= 1), b2 = wv::ballot(rows >= 2), b4 = wv::ballot(rows >= 4);
    int adr = nlim + wv::popc(b1 & lower) + wv::popc(b2 & lower) + 2 * wv::popc(b4 & lower);
    int total = nlim + wv::popc(b1) + wv::popc(b2) + 2 * wv::popc(b4);
    // contacts are admitted in order until the row cap is reached; a pyramid that does not fit is dropped whole
    bool fits = rows > 0 && adr + rows <= m.njmax;
    unsigned long long over = wv::ballot(rows > 0 && !fits);
    if (over) {
      // rare: re-run the prefix serially so that later, smaller pyramids are placed exactly as a serial sweep would
      wv::sync();
      if (L == 0) {
        int a = nlim;
        for (int c = 0; c < ncon; c++) {
          const real* C = S + l.con + CON_STRIDE * c;
          int g1 = I[l.i_cong1 + c], g2 = I[l.i_cong2 + c];
          int dim = T.geom_condim(g1) > T.geom_condim(g2) ? T.geom_condim(g1) : T.geom_condim(g2);
          int rws = dim == 1 ? 1 : 2 * ((dim < 3 ? dim : 3) - 1);
          int at = -1;
          if (C[CON_DIST] < C[CON_INCL] && a + rws <= m.njmax) { at = a; a += rws; }
          I[l.i_conadr + c] = at;
        }
        I[I_NEFC] = a;
      }
      wv::sync();
      warn |= 2;
      if (L < ncon) { adr = I[l.i_conadr + L]; fits = adr >= 0; }
      total = I[I_NEFC];
    }
    if (L < ncon) {
      I[l.i_conadr + L] = fits ? adr : -1;
      if (fits) for (int s = 0; s < rows; s++) I[l.i_rowid + adr + s] = L * 8 + s;
    }
    if (L == 0) { I[I_NEFC] = total; I[I_NLIM] = nlim; I[I_WARN] = warn; }
  }
  wv::sync();
  MJ_SUBSTAMP(ST_ROWS_ADDR)
#undef MJ_SUBSTAMP
  int nefc = I[I_NEFC];
  for (int r = L; r < nefc; r += 64) {
    real* Jr = S + l.J + JW * r;
    real* R = S + l.row + ROW_STRIDE * r;
    int id = I[l.i_rowid + r], rtree = -1;
    real pos, margin, diag, mu0 = 0;
    real sref[2], simp[5];      // solver parameters of the row, by value (a pointer into a local array would pin it to scratch memory)
    bool contact = id >= 0;
    // the row's dof chains: chain x = sparse-M row of dof lx, entries at ax .. ax + nx - 1 (descending dof ids)
    int a1 = 0, n1 = 0, a2 = 0, n2 = 0;
    real lim_val = 0, c_mu = 0, c_sgn = 0;
    int c_dim = 0;
    V3 c_n = v3(0, 0, 0), c_tk = v3(0, 0, 0), c_off1 = v3(0, 0, 0), c_off2 = v3(0, 0, 0);
    bool local = false;       // tree-local form (see JW)
    int adr0 = 0;             // first dof of the row's tree: the root of any of its chains
    // slot of dof i in a two-chain row.  Compact form: on chain 2 (primary) its position from the deepest dof, else
    // 8 + the position on chain 1
    auto slot = [&](int i) {
      if (local) return i - adr0;
      int dp = T.ddepth(i), t2 = n2 - 1 - dp;
      return (t2 >= 0 && T.colid(a2 + t2) == i) ? t2 : MAX_DOF_DEPTH + (n1 - 1 - dp);
    };
    if (!contact) {
      int it = -id - 1, j = it >> 1, side = (it & 1) ? 1 : -1;
      int dof = m.jnt_dofadr[j];
      lim_val = -side;
      rtree = T.dof_tree(dof);
      a2 = T.madr(dof); n2 = T.ddepth(dof) + 1;
      local = m.rowmap != 0;
      adr0 = T.colid(a2 + n2 - 1);
      pos = R[ROW_F];
      margin = m.jnt_margin[j];
      diag = m.dof_invweight0[dof];
      for (int k = 0; k < 2; k++) sref[k] = m.jnt_solref[2 * j + k];
      for (int k = 0; k < 5; k++) simp[k] = m.jnt_solimp[5 * j + k];
    } else {
      int c = id >> 3, sub = id & 7;
      const real* C = S + l.con + CON_STRIDE * c;
      int g1 = I[l.i_cong1 + c], g2 = I[l.i_cong2 + c];
      int b1 = T.geom_body(g1), b2 = T.geom_body(g2);
      int dim = T.geom_condim(g1) > T.geom_condim(g2) ? T.geom_condim(g1) : T.geom_condim(g2);
      V3 cp = ld3(C + CON_POS), n = ld3(C + CON_FRAME);
      real mu = C[CON_MU];     // friction[0] == friction[1]: both tangent directions share it
      mu0 = mu;
      int kt = 1 + (sub >> 1);
      real sgn = (sub & 1) ? -1.0 : 1.0;
      V3 tk = ld3(C + CON_FRAME + 3 * kt);
      int t1 = T.body_tree(b1), t2 = T.body_tree(b2);
      rtree = (t1 >= 0 && t2 >= 0 && t1 != t2) ? -2 : (t1 > t2 ? t1 : t2);   // -2: the row couples two trees
      V3 off1 = t1 >= 0 ? cp - ld3(S + l.com + 3 * t1) : v3(0, 0, 0);
      V3 off2 = t2 >= 0 ? cp - ld3(S + l.com + 3 * t2) : v3(0, 0, 0);
      int l1 = T.body_lastdof(b1), l2 = T.body_lastdof(b2);
      if (l1 >= 0) { a1 = T.madr(l1); n1 = T.ddepth(l1) + 1; }
      if (l2 >= 0) { a2 = T.madr(l2); n2 = T.ddepth(l2) + 1; }
      c_n = n; c_tk = tk; c_mu = mu; c_sgn = sgn; c_dim = dim; c_off1 = off1; c_off2 = off2;
      local = m.rowmap != 0 && rtree >= 0;
      if (n2 > 0) adr0 = T.colid(a2 + n2 - 1); else if (n1 > 0) adr0 = T.colid(a1 + n1 - 1);
      if (local) for (int k = 0; k < JW; k++) Jr[k] = 0;
      int p1 = 0, p2 = 0;
      // rows between two moving bodies walk the merged chains through LDS; one-chain rows are built in registers below
      int i1 = (n1 > 0 && n2 > 0) ? T.colid(a1) : -1, i2 = (n1 > 0 && n2 > 0) ? T.colid(a2) : -1;
      while (i1 >= 0 || i2 >= 0) {
        int i = i1 > i2 ? i1 : i2;
        V3 ca = ld3(S + l.cdof + 6 * i), cl = ld3(S + l.cdof + 6 * i + 3);
        V3 colv = v3(0, 0, 0);
        if (i2 == i) { colv = cl + cross(ca, off2); p2++; i2 = p2 < n2 ? T.colid(a2 + p2) : -1; }
        if (i1 == i) { colv = colv - (cl + cross(ca, off1)); p1++; i1 = p1 < n1 ? T.colid(a1 + p1) : -1; }
        real jn = dot(n, colv);
        Jr[slot(i)] = dim == 1 ? jn : jn + sgn * mu * dot(tk, colv);
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
      contact = dim > 1;
    }
    // One-chain rows (every limit row, every contact with a static geom): the row lives in registers, indexed by the
    // position along the chain (t = 0 deepest dof ... nc-1 tree root); all loops are unrolled to the chain-depth cap.
    const bool single = (n1 == 0 || n2 == 0);
    const int ac = n2 ? a2 : a1, nc = n2 ? n2 : n1;
    // the row's info word: chain code | (tree + 2) << CHAIN_BITS
    {
      int xp = nc ? T.colid(ac) : 0, dp = nc ? nc - 1 : 0;
      int chain = xp | (dp << 6);
      if (!single) chain |= ((T.colid(a1) + 1) << 9) | ((n1 - 1) << 16);
      I[l.i_rowinfo + r] = chain | ((rtree + 2) << CHAIN_BITS);
      if (nc == 0) Jr[0] = 0;      // (a row between two fixed bodies touches no dof)
      if (local && id < 0) for (int k = 0; k < JW; k++) Jr[k] = 0;     // (contact rows were cleared above)
    }
    real Bv[MAX_DOF_DEPTH];
    int cd[MAX_DOF_DEPTH];
    real vel = 0, ja = 0, jw = 0;
    if (single) {
#pragma unroll
      for (int t = 0; t < MAX_DOF_DEPTH; t++) { cd[t] = t < nc ? T.colid(ac + t) : 0; Bv[t] = 0; }
      if (id < 0) {
        Bv[0] = lim_val;
      } else {
        V3 offc = n2 ? c_off2 : c_off1;
#pragma unroll
        // (computed for all eight positions and zeroed past the chain's end -- positions there use dof 0 -- so that
        // neither this loop nor the dot products below run under a per-position lane mask)
        for (int t = 0; t < MAX_DOF_DEPTH; t++) {
          V3 ca = ld3(S + l.cdof + 6 * cd[t]), cl = ld3(S + l.cdof + 6 * cd[t] + 3);
          V3 colv = cl + cross(ca, offc);
          if (!n2) colv = v3(0, 0, 0) - colv;
          real jn = dot(c_n, colv);
          real val = c_dim == 1 ? jn : jn + c_sgn * c_mu * dot(c_tk, colv);
          Bv[t] = t < nc ? val : 0.0;
        }
      }
#pragma unroll
      for (int t = MAX_DOF_DEPTH - 1; t >= 0; t--) {      // ascending dof id
        vel += Bv[t] * S[l.qvel + cd[t]];
        ja += Bv[t] * S[l.qaccs + cd[t]];
        jw += Bv[t] * S[l.warm + cd[t]];
      }
    } else {
      int p1 = n1 - 1, p2 = n2 - 1;
      int i1 = p1 >= 0 ? T.colid(a1 + p1) : 1 << 20, i2 = p2 >= 0 ? T.colid(a2 + p2) : 1 << 20;
      while (p1 >= 0 || p2 >= 0) {
        int i = i1 < i2 ? i1 : i2;
        if (i1 == i) { p1--; i1 = p1 >= 0 ? T.colid(a1 + p1) : 1 << 20; }
        if (i2 == i) { p2--; i2 = p2 >= 0 ? T.colid(a2 + p2) : 1 << 20; }
        real jk = Jr[slot(i)];
        vel += jk * S[l.qvel + i];
        ja += jk * S[l.qaccs + i];
        jw += jk * S[l.warm + i];
      }
    }
    real imp = impedance(simp, pos, margin);
    real dmax = fmin(fmax(simp[1], MJ_MINIMP), MJ_MAXIMP);
    real timeconst = fmax(sref[0], 2 * m.timestep), dampratio = sref[1];
    real Kc = 1.0 / fmax(MJ_MINVAL, dmax * dmax * timeconst * timeconst * dampratio * dampratio);
    real Bc = 2.0 / fmax(MJ_MINVAL, dmax * timeconst);
    real Rr = fmax(MJ_MINVAL, (1 - imp) / imp * diag);
    if (contact) Rr = fmax(MJ_MINVAL, 2 * mu0 * mu0 * Rr);   // pyramid edges share 2*mu^2*R(first edge)
    real aref = -Bc * vel - Kc * imp * (pos - margin);
    real jar = jw - aref;
    real Dr = 1.0 / Rr;
    R[ROW_R] = Rr;
    R[ROW_B] = ja - aref;
    R[ROW_F] = jar < 0 ? -Dr * jar : 0.0;
    if (!project) {
      if (single) {
#pragma unroll
        for (int t = 0; t < MAX_DOF_DEPTH; t++) if (t < nc) Jr[local ? cd[t] - adr0 : t] = Bv[t];
      }
      continue;
    }
    // J <- J L^-1 restricted to the chains: descending over the union, each dof pushes its value to its ancestors
    real acc = 0;
    if (single) {
      // position t has depth nc-1-t; its ancestors are positions t+1 .. nc-1, the factor entries L[k][.] follow
      // the diagonal of row k in the sparse layout
#pragma unroll
      // (no predicates inside: positions past the chain's end hold values nobody reads, their updates flow only
      // further up, and the factor entries read for them are some other dof's, finite; a select per multiply-add cost
      // more than the arithmetic)
      for (int t = 0; t < MAX_DOF_DEPTH; t++) {
        real v = Bv[t];
        if (t < nc) acc += v * v * S[l.Dinv + cd[t]];
        int adr = T.madr(cd[t]);
#pragma unroll
        for (int sft = 1; sft < MAX_DOF_DEPTH - t; sft++) Bv[t + sft] -= v * S[l.LD + adr + sft];
      }
#pragma unroll
      for (int t = 0; t < MAX_DOF_DEPTH; t++) if (t < nc) Jr[local ? cd[t] - adr0 : t] = Bv[t];
    } else {
      int p1 = 0, p2 = 0;
      int i1 = n1 > 0 ? T.colid(a1) : -1, i2 = n2 > 0 ? T.colid(a2) : -1;
      while (i1 >= 0 || i2 >= 0) {
        int k = i1 > i2 ? i1 : i2;
        if (i1 == k) { p1++; i1 = p1 < n1 ? T.colid(a1 + p1) : -1; }
        if (i2 == k) { p2++; i2 = p2 < n2 ? T.colid(a2 + p2) : -1; }
        real v = Jr[slot(k)];
        acc += v * v * S[l.Dinv + k];
        if (v != 0.0) {
          int adr = T.madr(k), depth = T.ddepth(k);
          for (int t = 1; t <= depth; t++) Jr[slot(T.colid(adr + t))] -= v * S[l.LD + adr + t];
        }
      }
    }
    real aii = acc + Rr;
    R[ROW_ARII] = aii;
  }
  wv::sync();
}

// The register solver for the rare heavy step: at most two kinematic trees and 17..32 rows in one of them (an ant
// resting on many geoms).  Tree t's row k sits in lane 16 t + (k & 15) + 32 (k >> 4): rows 0..15 in the tree's
// row of 16 lanes, rows 16..31 in the row of 16 lanes two further up, which a model with two trees leaves idle.  A
// force change is broadcast inside its row of 16 by DPP and handed to the other half of the wave by one
// v_permlane32_swap per dword.  Same algorithm, guard and sweep count as the 16-row solver in stage_pgs; without
// this path such a copy -- the slowest of nearly every launch of 4096 -- ran the LDS-pipelined sweep at three times
// the cost per row step.  Returns u = B' f for the lane's dof.
template <int K>
__device__ __forceinline__ real wide_bcast(real v) {
  if constexpr (K < 16) return wv::half_to_all<false>(wv::bcast16<K>(v));
  else return wv::half_to_all<true>(wv::bcast16<K - 16>(v));
}
// (Inlined.  As a function of its own it cost the kernel a stack frame: its callee-saved registers and, while its
// arguments were a struct, the struct -- the only scratch memory of the step kernel.  Early in round 1 the inlined form
// drove the register allocation to 512 VGPRs; with the kernel as it is now it fits the 256 of two waves per SIMD.)
#ifndef MJRL_WIDE_INLINE
#define MJRL_WIDE_INLINE inline __attribute__((always_inline))
#endif
#define MJ_ROWS32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
                     X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
// (A real function, not inlined: inlined into the step kernel its 32 unrolled row steps drove the kernel's register
// allocation to 512 VGPRs plus scratch.  It takes plain values only -- a reference to the model or the layout would
// force those structs into memory.)
// (plain scalars, each in a register of its own: a struct by value travels through the stack -- scratch memory)
// (the sweep count goes in and comes back by value: a pointer to the caller's counter pinned that counter -- which every
// solver path increments once per sweep -- to scratch memory)
struct WideOut { real u; int iter; };
__device__ MJRL_WIDE_INLINE WideOut pgs_wide_registers(real* S, const int* I, int L, int cnt_w, int base_w, int tmax,
                                                        bool dof, int o_rowid, int o_row, int o_J, int o_Dinv, int iterations,
                                                        int adr0_in, real tolerance, real scale_in, int iter_in) {
  // o_*: LDS offsets (Lay); iterations: sweep cap; adr0_in: first dof of the lane's tree
  struct { int i_rowid, row, J, Dinv; } l = {o_rowid, o_row, o_J, o_Dinv};
  struct { int iterations; real tolerance; } m = {iterations, tolerance};
  const real scale = scale_in;
  int iter = wv::first_int(iter_in);
  tmax = wv::first_int(tmax);              // (arguments arrive in vector registers; these two are wave-uniform)
  m.iterations = wv::first_int(m.iterations);
  const int tree = (L >> 4) & 1;
  (void)tree;
  const int kme = (L & 15) + 16 * (L >> 5);
  const bool has_row = kme < cnt_w;
  const int myrow = has_row ? I[l.i_rowid + base_w + kme] : 0;
  real* Rm = S + l.row + ROW_STRIDE * myrow;
  real fi = has_row ? Rm[ROW_F] : 0.0, bi = has_row ? Rm[ROW_B] : 0.0, Ri = has_row ? Rm[ROW_R] : 0.0;
  const real aii = has_row ? Rm[ROW_ARII] : 1.0;
  const real ainv = 1.0 / aii;
  const int adr0 = adr0_in;
  real W[16], A[32];
#pragma unroll
  for (int d = 0; d < 16; d++)
    W[d] = has_row ? S[l.J + JW * myrow + d] * S[l.Dinv + adr0 + d] : 0.0;       // (slots past the tree's dofs hold 0)
#pragma unroll
  for (int k = 0; k < 32; k++) A[k] = 0;
#define MJ_ASTEP(KK)                                                                  \
    if (KK >= tmax) break;                                                            \
    {                                                                                 \
      const real* Bk = S + l.J + JW * (KK < cnt_w ? I[l.i_rowid + base_w + KK] : 0);  \
      real p0 = 0, p1 = 0, p2 = 0, p3 = 0;                                            \
      _Pragma("unroll")                                                               \
      for (int d = 0; d < 16; d += 4) {                                               \
        p0 += W[d] * Bk[d]; p1 += W[d + 1] * Bk[d + 1]; p2 += W[d + 2] * Bk[d + 2]; p3 += W[d + 3] * Bk[d + 3]; \
      }                                                                               \
      real acc = (p0 + p1) + (p2 + p3);                                               \
      if (KK >= cnt_w) acc = 0;                                                       \
      if (KK == kme) acc += Ri;                                                       \
      A[KK] = acc;                                                                    \
    }
  do { MJ_ROWS32(MJ_ASTEP) } while (0);
#undef MJ_ASTEP
  real r = bi;
#define MJ_RINIT(KK) if (KK >= tmax) break; r += A[KK] * wide_bcast<KK>(fi);
  do { MJ_ROWS32(MJ_RINIT) } while (0);
#undef MJ_RINIT
  {
    real cost = wv::rows4_sum(wv::sum16(0.5 * fi * (r + bi)));
    const bool cold = cost > 0;
    fi = cold ? 0.0 : fi;
    r = cold ? bi : r;
  }
  real sr = r * ainv;
#pragma unroll
  for (int k = 0; k < 32; k++) A[k] *= ainv;
  const real haii = 0.5 * aii;
  real f_start = fi, s_start = sr, f_prev = fi, s_prev = sr, c_prev = 0;
  bool pending = false, guarded = false;
  while (iter < m.iterations) {
    const int kme_s = wv::opaque_lane(kme), tmax_s = wv::opaque_uniform(tmax);
    f_start = fi; s_start = sr;
    real ns = -sr, nss = ns;                 // negated residual: a row step is max, broadcast, multiply-add (see stage_pgs)
    const real nf = -fi;
    // the stop test of the sweep before (see stage_pgs): the two lane masks are formed here, the branch on them comes
    // after the first two row steps, whose work covers the vector-to-scalar latency
    unsigned long long refused = 0ull, deciding = 1ull;
    if (pending) { refused = wv::ballot(c_prev > 1e-10); deciding = wv::ballot((-c_prev - 1e-8) * scale >= m.tolerance); }
#define MJ_FSTEP(KK)                                                                  \
      {                                                                               \
        real db = wide_bcast<KK>(fmax(ns, nf));                                       \
        if (kme_s == KK) nss = ns;                                                    \
        ns = __builtin_fma(-A[KK], db, ns);     /* one fused op on the chain (see stage_pgs) */ \
      }
    MJ_FSTEP(0) MJ_FSTEP(1)
    if (pending && (refused != 0ull || deciding == 0ull)) {
      const real improvement = wv::rows4_sum(wv::sum16(-c_prev));
      if (refused != 0ull || wv::ballot(improvement * scale < m.tolerance)) {
        if (refused != 0ull) { fi = f_prev; sr = s_prev; iter--; guarded = true; }
        else { sr = s_start; }
        pending = false;
        break;
      }
    }
    MJ_FSTEP(2) MJ_FSTEP(3) MJ_FSTEP(4) MJ_FSTEP(5) MJ_FSTEP(6) MJ_FSTEP(7) MJ_FSTEP(8) MJ_FSTEP(9)
    MJ_FSTEP(10) MJ_FSTEP(11) MJ_FSTEP(12) MJ_FSTEP(13) MJ_FSTEP(14) MJ_FSTEP(15)
    MJ_FSTEP(16) MJ_FSTEP(17) MJ_FSTEP(18) MJ_FSTEP(19)
    do {
      if (20 >= tmax_s) break;
      MJ_FSTEP(20) MJ_FSTEP(21) MJ_FSTEP(22) MJ_FSTEP(23)
      if (24 >= tmax_s) break;
      MJ_FSTEP(24) MJ_FSTEP(25) MJ_FSTEP(26) MJ_FSTEP(27)
      if (28 >= tmax_s) break;
      MJ_FSTEP(28) MJ_FSTEP(29) MJ_FSTEP(30) MJ_FSTEP(31)
    } while (0);
#undef MJ_FSTEP
    sr = -ns;
    const real ss = -nss;
    fi = fmax(f_start - ss, 0.0);
    const real dsweep = fi - f_start;
    c_prev = dsweep * dsweep * haii + dsweep * (ss * aii);
    f_prev = f_start; s_prev = s_start;
    pending = true;
    iter++;
  }
  if (pending && wv::ballot(c_prev > 1e-10)) { fi = f_prev; sr = s_prev; iter--; guarded = true; }
  while (guarded && iter < m.iterations) {
    real imp = 0;
    const int kme_g = wv::opaque_lane(kme);     // (per sweep, as above: 32 hoisted lane masks do not fit the scalar registers)
#define MJ_GSTEP(KK)                                                                  \
    if (KK < tmax) {                                                                  \
      real fn = fmax(fi - sr, 0.0);                                                   \
      real delta = fn - fi;                                                           \
      real change = delta * delta * haii + delta * (sr * aii);                        \
      bool act = kme_g == KK && has_row && !(change > 1e-10);                         \
      if (!act) { delta = 0; change = 0; fn = fi; }                                   \
      fi = fn;                                                                        \
      imp -= change;                                                                  \
      sr += A[KK] * wide_bcast<KK>(delta);                                            \
    }
    MJ_ROWS32(MJ_GSTEP)
#undef MJ_GSTEP
    iter++;
    if (wv::rows4_sum(wv::sum16(imp)) * scale < m.tolerance) break;
  }
  if (has_row) Rm[ROW_F] = fi;
  real u = 0;
#define MJ_USTEP(KK)                                                                  \
    if (KK >= tmax) break;                                                            \
    {                                                                                 \
      real fk = wide_bcast<KK>(fi);                                                   \
      const int rk = KK < cnt_w ? I[l.i_rowid + base_w + KK] : 0;                     \
      if (dof && KK < cnt_w) u += S[l.J + JW * rk + (L & 15)] * fk;                   \
    }
  do { MJ_ROWS32(MJ_USTEP) } while (0);
#undef MJ_USTEP
  WideOut out;
  out.u = u; out.iter = iter;
  return out;
}
#undef MJ_ROWS32

// Row lists for a copy in which some rows couple two kinematic trees (tree-row lane map): list[t * C + p] = the row
// tree t steps at position p of a sweep, -1 for none, C = njmax / ntree, written over the row-id array.  Rows keep
// their solver order inside every tree, and a coupling row takes the first position that is free in both its trees.
// Returns false (lists unusable) when a list would outgrow its C entries; `len` receives the sweep length.
__device__ inline bool pgs_coupled_schedule(const DevModel& m, const Lay& l, real* S, int L, int nefc, int& len) {
  int* I = (int*)(S + l.ints);
  const Tab T = make_tab(m, l, S);
  len = 0;
  if (m.ntree > 4 || m.ntree < 1) return false;
  const int C = m.njmax / m.ntree, n = wv::first_int(nefc);
  for (int k = L; k < m.ntree * C; k += 64) I[l.i_rowid + k] = -1;
  wv::sync();
  // entries used in each tree's list (wave-uniform), 16 bits per tree in one scalar: four counters selected by a
  // run-time tree index became an indexed array in scratch memory
  unsigned long long packed = 0ull;
  bool fits = true;
  auto used = [&](int t) { return (int)((packed >> (16 * t)) & 0xFFFFull); };
  auto set_used = [&](int t, int v) { packed = (packed & ~(0xFFFFull << (16 * t))) | ((unsigned long long)(v & 0xFFFF) << (16 * t)); };
  for (int i = 0; i < n; i++) {
    const int info = wv::first_int(I[l.i_rowinfo + i]);
    const int rt = (info >> CHAIN_BITS) - 2;
    if (rt >= 0) {
      const int p = used(rt);
      if (p < C) { if (L == 0) I[l.i_rowid + rt * C + p] = i; } else fits = false;
      set_used(rt, p + 1);
    } else {
      const int t1 = wv::first_int(T.dof_tree(info & 63)), t2 = wv::first_int(T.dof_tree(((info >> 9) & 127) - 1));
      const int p1 = used(t1), p2 = used(t2), p = p1 > p2 ? p1 : p2;
      if (p < C) { if (L == 0) { I[l.i_rowid + t1 * C + p] = i; I[l.i_rowid + t2 * C + p] = i; } } else fits = false;
      set_used(t1, p + 1);
      set_used(t2, p + 1);
    }
  }
  const int n0 = used(0), n1 = used(1), n2 = used(2), n3 = used(3);
  len = n0 > n1 ? n0 : n1;
  len = n2 > len ? n2 : len;
  len = n3 > len ? n3 : len;
  wv::sync();
  return fits;
}


// lane K & 15 of the caller's row of 16 lanes: of `lo` for K < 16, of `hi` for K >= 16 (two values per lane)
template <int K>
__device__ __forceinline__ real tall_bcast(real lo, real hi) {
  if constexpr (K < 16) return wv::bcast16<K>(lo);
  else return wv::bcast16<K - 16>(hi);
}

// The register solver for 17..32 rows in a tree of a model with MORE than two trees (the 4-agent arena: an ant pressed
// onto the floor; some 25 copies of every launch of 4096, and with the schedule sweep the longest waves of nearly every
// launch -- 170 us against a mean of 68).  All four rows of 16 lanes are taken by trees here, so a lane owns TWO rows of
// its tree: lane k of the tree's 16 keeps rows k and k + 16 -- their forces, their rows of AR = B D^-1 B' + diag(R)
// (2 x 32 registers pairs) and their residuals.  A row step is the 16-row solver's (stage_pgs): max on the negated
// residual of the row's slot, one DPP broadcast inside the tree's row of 16, one fused multiply-add per slot; the same
// late stop test, the same guard, the same sweep count.  It needs some 330 vector registers: such a model's LDS image
// holds a CU to one wave per SIMD anyway (4 copies of 38 KB), so the registers are there.
#ifndef MJRL_TALL_INLINE
#if defined(MJRL_SPEC) || !defined(__HIPCC__)
#define MJRL_TALL_INLINE inline __attribute__((always_inline))
#else
#define MJRL_TALL_INLINE __attribute__((noinline))      // (generic GPU kernel: a register allocation of its own)
#endif
#endif
struct TallOut { real u; int iter; };
#define MJ_ROWS32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
                     X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
__device__ MJRL_TALL_INLINE TallOut pgs_tall_registers(real* S, const int* I, int L, int cnt_in, int base_in, int tmax_in,
                                                       bool dof, int o_rowid, int o_row, int o_J, int o_Dinv, int iterations_in,
                                                       int adr0, int ntree, real tolerance, real scale, int iter_in) {
  int iter = wv::first_int(iter_in);
  const int tmax = wv::first_int(tmax_in), iterations = wv::first_int(iterations_in);
  const int cnt_my = cnt_in, base_my = base_in;
  const int kme = L & 15;
  const bool has0 = kme < cnt_my, has1 = kme + 16 < cnt_my;
  const int row0 = has0 ? I[o_rowid + base_my + kme] : 0, row1 = has1 ? I[o_rowid + base_my + kme + 16] : 0;
  real* Rm0 = S + o_row + ROW_STRIDE * row0;
  real* Rm1 = S + o_row + ROW_STRIDE * row1;
  real f0 = has0 ? Rm0[ROW_F] : 0.0, f1 = has1 ? Rm1[ROW_F] : 0.0;
  const real b0 = has0 ? Rm0[ROW_B] : 0.0, b1 = has1 ? Rm1[ROW_B] : 0.0;
  const real R0 = has0 ? Rm0[ROW_R] : 0.0, R1 = has1 ? Rm1[ROW_R] : 0.0;
  const real aii0 = has0 ? Rm0[ROW_ARII] : 1.0, aii1 = has1 ? Rm1[ROW_ARII] : 1.0;
  const real ainv0 = 1.0 / aii0, ainv1 = 1.0 / aii1;
  real W0[16], W1[16], A0[32], A1[32];
#pragma unroll
  for (int d = 0; d < 16; d++) {
    const real di = S[o_Dinv + adr0 + d];                                        // (slots past the tree's dofs hold 0 in J)
    W0[d] = has0 ? S[o_J + JW * row0 + d] * di : 0.0;
    W1[d] = has1 ? S[o_J + JW * row1 + d] * di : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 32; k++) { A0[k] = 0; A1[k] = 0; }
  // rows of AR: the tree's KK-th row against the lane's two rows (its id over DPP from the lane that owns it)
#define MJ_ASTEP(KK)                                                                  \
    if (KK >= tmax) break;                                                            \
    {                                                                                 \
      const int rk = KK < 16 ? wv::bcast16i<(KK & 15)>(row0) : wv::bcast16i<(KK & 15)>(row1); \
      const real* Bk = S + o_J + JW * rk;                                             \
      real p0 = 0, p1 = 0, p2 = 0, p3 = 0, q0 = 0, q1 = 0, q2 = 0, q3 = 0;            \
      _Pragma("unroll")                                                               \
      for (int d = 0; d < 16; d += 4) {                                               \
        const real e0 = Bk[d], e1 = Bk[d + 1], e2 = Bk[d + 2], e3 = Bk[d + 3];        \
        p0 += W0[d] * e0; p1 += W0[d + 1] * e1; p2 += W0[d + 2] * e2; p3 += W0[d + 3] * e3; \
        q0 += W1[d] * e0; q1 += W1[d + 1] * e1; q2 += W1[d + 2] * e2; q3 += W1[d + 3] * e3; \
      }                                                                               \
      real acc0 = (p0 + p1) + (p2 + p3), acc1 = (q0 + q1) + (q2 + q3);                \
      if (KK >= cnt_my) { acc0 = 0; acc1 = 0; }                                       \
      if (KK == kme) acc0 += R0;                                                      \
      if (KK == kme + 16) acc1 += R1;                                                 \
      A0[KK] = acc0; A1[KK] = acc1;                                                   \
    }
  do { MJ_ROWS32(MJ_ASTEP) } while (0);
#undef MJ_ASTEP
  real r0 = b0, r1 = b1;
#define MJ_RINIT(KK) if (KK >= tmax) break; { const real fk = tall_bcast<KK>(f0, f1); r0 += A0[KK] * fk; r1 += A1[KK] * fk; }
  do { MJ_ROWS32(MJ_RINIT) } while (0);
#undef MJ_RINIT
  {      // warm start: keep last step's forces only if they beat f = 0; cost(f) = 1/2 f'(r + b)
    const real cost = wv::rows_sum(wv::sum16(0.5 * f0 * (r0 + b0) + 0.5 * f1 * (r1 + b1)), ntree);
    const bool cold = cost > 0;
    f0 = cold ? 0.0 : f0; f1 = cold ? 0.0 : f1;
    r0 = cold ? b0 : r0; r1 = cold ? b1 : r1;
  }
  real s0 = r0 * ainv0, s1 = r1 * ainv1;             // scaled residuals (stage_pgs)
#pragma unroll
  for (int k = 0; k < 32; k++) { A0[k] *= ainv0; A1[k] *= ainv1; }
  const real haii0 = 0.5 * aii0, haii1 = 0.5 * aii1;
  real f0_start = f0, f1_start = f1, s0_start = s0, s1_start = s1;
  real f0_prev = f0, f1_prev = f1, s0_prev = s0, s1_prev = s1;
  real c0_prev = 0, c1_prev = 0;
  bool pending = false, guarded = false;
  while (iter < iterations) {
    const int kme_s = wv::opaque_lane(kme), tmax_s = wv::opaque_uniform(tmax);
    f0_start = f0; f1_start = f1; s0_start = s0; s1_start = s1;
    real ns0 = -s0, ns1 = -s1, nss0 = ns0, nss1 = ns1;
    const real nf0 = -f0, nf1 = -f1;
    // the stop test of the sweep before, decided in the shadow of this sweep's first two row steps (stage_pgs); one row
    // whose own cost decrease exceeds the threshold by more than the other rows could take back (64 x 1e-10) decides
    unsigned long long refused = 0ull, deciding = 1ull;
    if (pending) {
      refused = wv::ballot(c0_prev > 1e-10 || c1_prev > 1e-10);
      deciding = wv::ballot((fmax(-c0_prev, -c1_prev) - 1e-8) * scale >= tolerance);
    }
#define MJ_FSTEP(KK)                                                                  \
      {                                                                               \
        const real db = wv::bcast16<(KK & 15)>(KK < 16 ? fmax(ns0, nf0) : fmax(ns1, nf1)); \
        if (kme_s == (KK & 15)) { if (KK < 16) nss0 = ns0; else nss1 = ns1; }         \
        ns0 = __builtin_fma(-A0[KK], db, ns0);                                        \
        ns1 = __builtin_fma(-A1[KK], db, ns1);                                        \
      }
    MJ_FSTEP(0) MJ_FSTEP(1)
    if (pending && (refused != 0ull || deciding == 0ull)) {
      const real improvement = wv::rows_sum(wv::sum16(-c0_prev - c1_prev), ntree);
      if (refused != 0ull || wv::ballot(improvement * scale < tolerance)) {
        if (refused != 0ull) { f0 = f0_prev; f1 = f1_prev; s0 = s0_prev; s1 = s1_prev; iter--; guarded = true; }
        else { s0 = s0_start; s1 = s1_start; }
        pending = false;
        break;
      }
    }
    MJ_FSTEP(2) MJ_FSTEP(3) MJ_FSTEP(4) MJ_FSTEP(5) MJ_FSTEP(6) MJ_FSTEP(7) MJ_FSTEP(8) MJ_FSTEP(9)
    MJ_FSTEP(10) MJ_FSTEP(11) MJ_FSTEP(12) MJ_FSTEP(13) MJ_FSTEP(14) MJ_FSTEP(15) MJ_FSTEP(16)
    do {
      if (17 >= tmax_s) break;
      MJ_FSTEP(17) MJ_FSTEP(18) MJ_FSTEP(19)
      if (20 >= tmax_s) break;
      MJ_FSTEP(20) MJ_FSTEP(21) MJ_FSTEP(22) MJ_FSTEP(23)
      if (24 >= tmax_s) break;
      MJ_FSTEP(24) MJ_FSTEP(25) MJ_FSTEP(26) MJ_FSTEP(27)
      if (28 >= tmax_s) break;
      MJ_FSTEP(28) MJ_FSTEP(29) MJ_FSTEP(30) MJ_FSTEP(31)
    } while (0);
#undef MJ_FSTEP
    s0 = -ns0; s1 = -ns1;
    const real ss0 = -nss0, ss1 = -nss1;
    f0 = fmax(f0_start - ss0, 0.0); f1 = fmax(f1_start - ss1, 0.0);
    const real d0 = f0 - f0_start, d1 = f1 - f1_start;
    c0_prev = d0 * d0 * haii0 + d0 * (ss0 * aii0);
    c1_prev = d1 * d1 * haii1 + d1 * (ss1 * aii1);
    f0_prev = f0_start; f1_prev = f1_start; s0_prev = s0_start; s1_prev = s1_start;
    pending = true;
    iter++;
  }
  if (pending && wv::ballot(c0_prev > 1e-10 || c1_prev > 1e-10)) {      // the sweep cap was reached and the last sweep has a refused step
    f0 = f0_prev; f1 = f1_prev; s0 = s0_prev; s1 = s1_prev; iter--; guarded = true;
  }
  while (guarded && iter < iterations) {
    real imp = 0;
    const int kme_g = wv::opaque_lane(kme);
#define MJ_GSTEP(KK)                                                                  \
    if (KK < tmax) {                                                                  \
      const real fo = KK < 16 ? f0 : f1, so = KK < 16 ? s0 : s1;                      \
      const real ao = KK < 16 ? aii0 : aii1, ho = KK < 16 ? haii0 : haii1;            \
      real fn = fmax(fo - so, 0.0);                                                   \
      real delta = fn - fo;                                                           \
      real change = delta * delta * ho + delta * (so * ao);                           \
      const bool act = kme_g == (KK & 15) && (KK < 16 ? has0 : has1) && !(change > 1e-10); \
      if (!act) { delta = 0; change = 0; fn = fo; }                                   \
      if (KK < 16) f0 = fn; else f1 = fn;                                             \
      imp -= change;                                                                  \
      const real db = wv::bcast16<(KK & 15)>(delta);                                  \
      s0 += A0[KK] * db;                                                              \
      s1 += A1[KK] * db;                                                              \
    }
    MJ_ROWS32(MJ_GSTEP)
#undef MJ_GSTEP
    iter++;
    if (wv::rows_sum(wv::sum16(imp), ntree) * scale < tolerance) break;
  }
  if (has0) Rm0[ROW_F] = f0;
  if (has1) Rm1[ROW_F] = f1;
  real u = 0;       // u = B' f for the lane's dof
#define MJ_USTEP(KK)                                                                  \
    if (KK >= tmax) break;                                                            \
    {                                                                                 \
      const real fk = tall_bcast<KK>(f0, f1);                                         \
      const int rk = KK < 16 ? wv::bcast16i<(KK & 15)>(row0) : wv::bcast16i<(KK & 15)>(row1); \
      if (dof && KK < cnt_my) u += S[o_J + JW * rk + kme] * fk;                       \
    }
  do { MJ_ROWS32(MJ_USTEP) } while (0);
#undef MJ_USTEP
  TallOut out;
  out.u = u; out.iter = iter;
  return out;
}
#undef MJ_ROWS32

// Arguments of the register solver for a copy on a row schedule (pgs_coupled_schedule: every tree's list of rows, coupling
// rows at the same position of both their trees' lists).
struct SchedArgs {
  int o_rowid, o_rowinfo, o_row, o_J, o_tab;   // LDS offsets (Lay)
  int o_Dinv, adr0;                            // 1 / D of the factorised inertia matrix, first dof of the lane's tree
  int base, len, iterations;                   // the lane's list, sweep length, sweep cap
  int tab_dtree, tab_bytes;                    // dof -> tree table (Tab)
  int depth;                                   // of the lane's dof (RowK)
  unsigned long long below;
  real tolerance, scale, dinv, u;
};
// The sweep of a copy on a row schedule (lane k of a tree's 16 owns position k of the tree's list) in RESIDUAL form --
// the 16-row register solver of stage_pgs, generalised to rows that couple two trees.  (Round 2 and most of round 3 ran
// these copies in u-form from registers -- pgs_schedule_registers: the lane's coefficient at every position, the rows'
// records broadcast by DPP, a 16-lane reduction per position: 45 instructions per position.)  Such a row sits at the same
// position of both its trees' lists and is kept twice, once per tree: each copy owns the part of the row's residual
// that comes through its tree's dofs, r_t = (B_t D_t^-1 B_t' f), with R f + b added in the lower-numbered tree's copy.
// A step on a coupling position adds the two parts through the LDS crossbar (a + b == b + a: both copies get the same
// bits), takes the step in both copies at once, and every tree updates its own rows' residuals with its own column of
// AR.  Everything else -- one max, one DPP broadcast and one fused multiply-add per row step, the cost changes from
// captured residuals once per sweep and one sweep late, the guarded sweeps -- is the 16-row solver's, so a copy with an
// agent-against-agent contact costs a dozen instructions per position instead of 45 (those waves took 140-180 us on the
// 4-agent arena, up to 320: the waves its launches waited for).
// AR's columns for a coupling position cannot be read from the row's J block (it is stored along its two chains, not in
// the tree's slots): they are formed from the lanes' own coefficients (lane = dof, `bid`), broadcast one dof at a time.
// (NP = 16: one position per lane; NP = 32: two, lane k owns positions k and k + 16 -- some 400 registers, for images
// that hold a CU to one wave per SIMD anyway, like pgs_tall_registers)
#define MJ_POS32(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
                    X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
template <int NP>
__device__ inline __attribute__((always_inline)) real pgs_schedule_residual(real* S, const int* I, int L, bool dof, SchedArgs w, int ntree, int& iter_io) {
  constexpr int RP = NP / 16;     // records (positions) per lane
  const int k = L & 15, mytree = L >> 4;
  const bool mine = w.len > 0;
  const int len = wv::first_int(w.len), iterations = wv::first_int(w.iterations);
  int iter = wv::first_int(iter_io);
  const unsigned char* t8 = (const unsigned char*)(S + w.o_tab);
  auto dof_tree = [&](int d) { return w.tab_bytes ? (int)t8[w.tab_dtree + d] : (int)((const unsigned short*)t8)[w.tab_dtree + d]; };
  auto entry = [&](int p) { return (mine && p < len) ? I[w.o_rowid + w.base + p] : -1; };
  auto coef = [&](int r, int info) -> real {
    const int rt = (info >> CHAIN_BITS) - 2;
    const int sl = rt >= 0 ? (rt == mytree ? k : -1) : row_slot(info, w.below, w.depth);
    return (dof && sl >= 0) ? S[w.o_J + JW * r + sl] : 0.0;
  };
  // the lane's (= its dof's) coefficient at position p of its tree's list
  auto bid = [&](int p) -> real {
    const int e = entry(p);
    return e >= 0 ? coef(e, I[w.o_rowinfo + (e >= 0 ? e : 0)]) : 0.0;
  };
  // the records of the lane's own positions
  bool has[RP];
  int row[RP], has_i[RP], partner[RP];      // partner: first lane of the other tree of a coupling row, else -1
  real fi[RP], bi[RP], Ri[RP], aii[RP], ainv[RP], haii[RP];
  real once[RP];                            // 0 in the second copy of a coupling row: its cost change, R f and b count once
  real* Rm[RP];
#pragma unroll
  for (int j = 0; j < RP; j++) {
    const int e = entry(k + 16 * j);
    has[j] = e >= 0; has_i[j] = has[j] ? 1 : 0;
    row[j] = has[j] ? e : 0;
    Rm[j] = S + w.o_row + ROW_STRIDE * row[j];
    fi[j] = has[j] ? Rm[j][ROW_F] : 0.0; bi[j] = has[j] ? Rm[j][ROW_B] : 0.0; Ri[j] = has[j] ? Rm[j][ROW_R] : 0.0;
    aii[j] = has[j] ? Rm[j][ROW_ARII] : 1.0;
    ainv[j] = 1.0 / aii[j];
    haii[j] = 0.5 * aii[j];
    partner[j] = -1;
    once[j] = has[j] ? 1.0 : 0.0;
    const int info = I[w.o_rowinfo + row[j]];
    if (has[j] && (info >> CHAIN_BITS) == 0) {
      const int t1 = dof_tree(info & 63), t2 = dof_tree(((info >> 9) & 127) - 1);
      partner[j] = 16 * (t1 == mytree ? t2 : t1);
      if (mytree != (t1 < t2 ? t1 : t2)) { once[j] = 0.0; bi[j] = 0.0; Ri[j] = 0.0; }
    }
  }
  unsigned anyc = 0u;             // positions at which some tree has a coupling row
#pragma unroll
  for (int j = 0; j < RP; j++) {
    const unsigned long long c = wv::ballot(partner[j] >= 0);
    anyc |= (unsigned)((c | (c >> 16) | (c >> 32) | (c >> 48)) & 0xFFFFull) << (16 * j);
  }
  // AR, one of the lane's own rows at a time (both at once is 64 more registers at the solver's widest point):
  // W = (the own row's coefficients in its tree's 16 slots) x D^-1 -- a tree-local row's are its J block, a coupling
  // row's come over DPP from the dof lanes --, then column P of AR against it, the column's coefficients streamed
  real A[RP][NP];
#pragma unroll
  for (int j = 0; j < RP; j++) {
    real W[16];
#pragma unroll
    for (int d = 0; d < 16; d++)
      W[d] = (has[j] && partner[j] < 0) ? S[w.o_J + JW * row[j] + d] * S[w.o_Dinv + w.adr0 + d] : 0.0;
    // (every chain of per-position tests works on its own copies of `len` and `anyc`, their origin hidden: shared, the
    // compiler forms all 2 NP tests once, ahead of the first chain, and keeps them -- in spilled scalar registers.  Not
    // in the two-position form: it has no vector register to spare for the detour, and spills to scratch.)
    const int len_w = NP == 16 ? wv::opaque_uniform(len) : len;
    const unsigned anyc_w = NP == 16 ? (unsigned)wv::opaque_uniform((int)anyc) : anyc;
#define MJ_WSTEP(P)                                                                   \
    if (P >= NP || P >= len_w) break;                                                 \
    if (((anyc_w >> P) & 1u) && (P < NP ? P : 0) / 16 == j) {                           \
      const real mine_p = bid(P);                                                     \
      _Pragma("unroll")                                                               \
      for (int d = 0; d < 16; d++) {                                                  \
        const real x = wv::bcast16_var(mine_p, d);                                    \
        if (k == (P & 15) && partner[j] >= 0) W[d] = x * S[w.o_Dinv + w.adr0 + d];    \
      }                                                                               \
    }
    do { MJ_POS32(MJ_WSTEP) } while (0);
#undef MJ_WSTEP
#pragma unroll
    for (int q = 0; q < NP; q++) A[j][q] = 0;
    const int len_a = NP == 16 ? wv::opaque_uniform(len) : len;
    const unsigned anyc_a = NP == 16 ? (unsigned)wv::opaque_uniform((int)anyc) : anyc;
#define MJ_ASTEP(P)                                                                   \
    if (P >= NP || P >= len_a) break;                                                 \
    {                                                                                 \
      constexpr int PP = P < NP ? P : 0, PJ = PP / 16;                                \
      real p0 = 0, p1 = 0, p2 = 0, p3 = 0;                                            \
      if ((anyc_a >> P) & 1u) {                                                         \
        const real mine_p = bid(P);                                                   \
        _Pragma("unroll")                                                             \
        for (int d = 0; d < 16; d += 4) {                                             \
          p0 += W[d] * wv::bcast16_var(mine_p, d); p1 += W[d + 1] * wv::bcast16_var(mine_p, d + 1); \
          p2 += W[d + 2] * wv::bcast16_var(mine_p, d + 2); p3 += W[d + 3] * wv::bcast16_var(mine_p, d + 3); \
        }                                                                             \
      } else {                                                                        \
        const real* Bk = S + w.o_J + JW * wv::bcast16i<(P & 15)>(row[PJ]);            \
        _Pragma("unroll")                                                             \
        for (int d = 0; d < 16; d += 4) {                                             \
          p0 += W[d] * Bk[d]; p1 += W[d + 1] * Bk[d + 1]; p2 += W[d + 2] * Bk[d + 2]; p3 += W[d + 3] * Bk[d + 3]; \
        }                                                                             \
        if (!wv::bcast16i<(P & 15)>(has_i[PJ])) { p0 = 0; p1 = 0; p2 = 0; p3 = 0; }   \
      }                                                                               \
      real acc = (p0 + p1) + (p2 + p3);                                               \
      if (k == (P & 15) && j == PJ) acc += Ri[j];                                     \
      A[j][PP] = acc;                                                                 \
    }
    do { MJ_POS32(MJ_ASTEP) } while (0);
#undef MJ_ASTEP
  }
  real sr[RP];
#pragma unroll
  for (int j = 0; j < RP; j++) sr[j] = bi[j];
  const int len_r = NP == 16 ? wv::opaque_uniform(len) : len;
#define MJ_RINIT(P)                                                                   \
  if (P >= NP || P >= len_r) break;                                                   \
  {                                                                                   \
    const real fk = wv::bcast16<(P & 15)>(fi[(P < NP ? P : 0) / 16]);                 \
    _Pragma("unroll")                                                                 \
    for (int j = 0; j < RP; j++) sr[j] += A[j][P < NP ? P : 0] * fk;                  \
  }
  do { MJ_POS32(MJ_RINIT) } while (0);
#undef MJ_RINIT
  // scaled residuals s = r / AR_kk and AR's rows scaled likewise (as in the 16-row solver)
#pragma unroll
  for (int j = 0; j < RP; j++) {
    sr[j] *= ainv[j];
#pragma unroll
    for (int q = 0; q < NP; q++) A[j][q] *= ainv[j];
  }
  const real scale = w.scale, tolerance = w.tolerance;
  // the row's whole (scaled, negated) residual at a coupling position: this copy's part plus the other tree's
  // (through the LDS crossbar; reading every tree's lane into scalar registers and picking the partner's -- a dozen
  // instructions, no LDS round trip -- measured no faster: 13.75 against 13.80 M env-steps/s on the 4-agent arena)
  // (anyc_s, part_s: per-sweep copies of anyc and partner whose origin the optimiser cannot see -- it would hoist one
  // 64-bit mask per position and per test out of the sweeps, and they do not fit the scalar registers)
#define MJ_WHOLE(P, v)                                                                \
  if ((anyc_s >> P) & 1u) {                                                           \
    const int pl = part_s[(P < NP ? P : 0) / 16];                                     \
    const real other = wv::shfl(v, pl >= 0 ? pl + k : L);                             \
    if (pl >= 0) v += other;                                                          \
  }
  real f_start[RP], s_start[RP], f_prev[RP], s_prev[RP], c_prev[RP];
#pragma unroll
  for (int j = 0; j < RP; j++) { f_start[j] = fi[j]; s_start[j] = sr[j]; f_prev[j] = fi[j]; s_prev[j] = sr[j]; c_prev[j] = 0; }
  bool pending = false, guarded = false;
  auto any_refused = [&]() {
    bool b = c_prev[0] > 1e-10;
    if constexpr (RP > 1) b = b || c_prev[RP - 1] > 1e-10;
    return wv::ballot(b);
  };
  auto cost_sum = [&]() {
    real c = -c_prev[0];
    if constexpr (RP > 1) c -= c_prev[RP - 1];
    return wv::rows_sum(wv::sum16(c), ntree);
  };
  while (iter < iterations) {
    const int k_s = wv::opaque_lane(k), len_s = wv::opaque_uniform(len);
    const unsigned anyc_s = (unsigned)wv::opaque_uniform((int)anyc);
    int part_s[RP];
#pragma unroll
    for (int j = 0; j < RP; j++) part_s[j] = wv::opaque_lane(partner[j]);
    real ns[RP], nss[RP], nf[RP];
#pragma unroll
    for (int j = 0; j < RP; j++) { f_start[j] = fi[j]; s_start[j] = sr[j]; ns[j] = -sr[j]; nss[j] = ns[j]; nf[j] = -fi[j]; }
    unsigned long long refused = 0ull, deciding = 1ull;
    if (pending) {
      refused = any_refused();
      bool dec = (-c_prev[0] - 1e-8) * scale >= tolerance;
      if constexpr (RP > 1) dec = dec || (-c_prev[RP - 1] - 1e-8) * scale >= tolerance;
      deciding = wv::ballot(dec);
    }
#define MJ_FSTEP(P)                                                                   \
    {                                                                                 \
      constexpr int PP = P < NP ? P : 0, PJ = PP / 16;                                \
      real nsx = ns[PJ];                                                              \
      MJ_WHOLE(P, nsx)                                                                \
      const real db = wv::bcast16<(P & 15)>(fmax(nsx, nf[PJ]));                       \
      if (k_s == (P & 15)) nss[PJ] = nsx;                                             \
      _Pragma("unroll")                                                               \
      for (int j = 0; j < RP; j++) ns[j] = __builtin_fma(-A[j][PP], db, ns[j]);       \
    }
    MJ_FSTEP(0) MJ_FSTEP(1)
    if (pending && (refused != 0ull || deciding == 0ull)) {
      const real improvement = cost_sum();
      if (refused != 0ull || wv::ballot(improvement * scale < tolerance)) {
        if (refused != 0ull) {                                  // redo that sweep, guarded
#pragma unroll
          for (int j = 0; j < RP; j++) { fi[j] = f_prev[j]; sr[j] = s_prev[j]; }
          iter--; guarded = true;
        } else {                                                // it had converged
#pragma unroll
          for (int j = 0; j < RP; j++) sr[j] = s_start[j];
        }
        pending = false;
        break;
      }
    }
    do {
      if (2 >= len_s) break;
      MJ_FSTEP(2) MJ_FSTEP(3)
      if (4 >= len_s) break;
      MJ_FSTEP(4) MJ_FSTEP(5) MJ_FSTEP(6) MJ_FSTEP(7)
      if (8 >= len_s) break;
      MJ_FSTEP(8) MJ_FSTEP(9) MJ_FSTEP(10) MJ_FSTEP(11)
      if (12 >= len_s) break;
      MJ_FSTEP(12) MJ_FSTEP(13) MJ_FSTEP(14) MJ_FSTEP(15)
      if (NP <= 16 || 16 >= len_s) break;
      MJ_FSTEP(16) MJ_FSTEP(17) MJ_FSTEP(18) MJ_FSTEP(19)
      if (20 >= len_s) break;
      MJ_FSTEP(20) MJ_FSTEP(21) MJ_FSTEP(22) MJ_FSTEP(23)
      if (24 >= len_s) break;
      MJ_FSTEP(24) MJ_FSTEP(25) MJ_FSTEP(26) MJ_FSTEP(27)
      if (28 >= len_s) break;
      MJ_FSTEP(28) MJ_FSTEP(29) MJ_FSTEP(30) MJ_FSTEP(31)
    } while (0);
#undef MJ_FSTEP
#pragma unroll
    for (int j = 0; j < RP; j++) {
      sr[j] = -ns[j];
      const real ss = -nss[j];
      fi[j] = fmax(f_start[j] - ss, 0.0);
      const real dsweep = fi[j] - f_start[j];
      c_prev[j] = (dsweep * dsweep * haii[j] + dsweep * (ss * aii[j])) * once[j];
      f_prev[j] = f_start[j]; s_prev[j] = s_start[j];
    }
    pending = true;
    iter++;
  }
  if (pending && any_refused()) {       // the sweep cap was reached and the last sweep has a refused step
#pragma unroll
    for (int j = 0; j < RP; j++) { fi[j] = f_prev[j]; sr[j] = s_prev[j]; }
    iter--; guarded = true;
  }
  while (guarded && iter < iterations) {
    real imp = 0;
    const int k_g = wv::opaque_lane(k);
    const unsigned anyc_s = (unsigned)wv::opaque_uniform((int)anyc);
    int part_s[RP];
#pragma unroll
    for (int j = 0; j < RP; j++) part_s[j] = wv::opaque_lane(partner[j]);
    const int len_g = wv::opaque_uniform(len);
#define MJ_GSTEP(P)                                                                   \
    if (P >= NP || P >= len_g) break;                                                 \
    {                                                                                 \
      constexpr int PP = P < NP ? P : 0, PJ = PP / 16;                                \
      real srx = sr[PJ];                                                              \
      MJ_WHOLE(P, srx)                                                                \
      real fn = fmax(fi[PJ] - srx, 0.0);                                              \
      real delta = fn - fi[PJ];                                                       \
      real change = delta * delta * haii[PJ] + delta * (srx * aii[PJ]);               \
      const bool act = k_g == (P & 15) && has[PJ] && !(change > 1e-10);               \
      if (!act) { delta = 0; change = 0; fn = fi[PJ]; }                               \
      fi[PJ] = fn;                                                                    \
      imp -= change * once[PJ];                                                       \
      const real dk = wv::bcast16<(P & 15)>(delta);                                   \
      _Pragma("unroll")                                                               \
      for (int j = 0; j < RP; j++) sr[j] += A[j][PP] * dk;                            \
    }
    do { MJ_POS32(MJ_GSTEP) } while (0);
#undef MJ_GSTEP
    iter++;
    if (wv::rows_sum(wv::sum16(imp), ntree) * scale < tolerance) break;
  }
#undef MJ_WHOLE
#pragma unroll
  for (int j = 0; j < RP; j++)
    if (has[j] && once[j] != 0.0) Rm[j][ROW_F] = fi[j];
  // u = B' f for the lane's dof (its coefficients fetched again: kept across the sweeps they would cost NP registers)
  real u = 0;
  const int len_u = NP == 16 ? wv::opaque_uniform(len) : len;
#define MJ_USTEP(P)                                                                   \
  if (P >= NP || P >= len_u) break;                                                   \
  {                                                                                   \
    u += bid(P) * wv::bcast16<(P & 15)>(fi[(P < NP ? P : 0) / 16]);                   \
  }
  do { MJ_POS32(MJ_USTEP) } while (0);
#undef MJ_USTEP
  iter_io = iter;
  return u;
}
#undef MJ_POS32

// The two-position form as the kernels call it: inlined into a model-specialised kernel (and the CPU emulation), a
// function with a register allocation of its own in the generic GPU kernel, like pgs_tall_registers there.
#if defined(MJRL_SPEC) || !defined(__HIPCC__)
#define MJRL_SCHED32_INLINE inline __attribute__((always_inline))
#else
#define MJRL_SCHED32_INLINE __attribute__((noinline))
#endif
__device__ MJRL_SCHED32_INLINE real pgs_schedule_residual32(real* S, const int* I, int L, bool dof, const SchedArgs& w, int ntree,
                                                            int iter_in, int* iter_out) {
  int it = iter_in;
  const real u = pgs_schedule_residual<32>(S, I, L, dof, w, ntree, it);
  *iter_out = it;
  return u;
}

// a model whose LDS image holds a CU to one wave per SIMD anyway: the forms that need more than 256 registers are free
__host__ __device__ __forceinline__ bool pgs_roomy(const DevModel& m, const Lay& l) {
  return m.ntree > 2 && (size_t)l.total * sizeof(real) > 32 * 1024;
}

// projected Gauss-Seidel on the dual  min 1/2 f'(A+R)f + f'b, f >= 0, with A = B D^-1 B' never formed: the lane that
// owns dof d carries u_d = (B' f)_d, a row's residual is one reduction over its tree's lanes, its update one
// multiply-add.  In the tree-row lane map a constraint row that touches one kinematic tree only involves that tree's
// row of 16 lanes, so the trees sweep their own rows side by side (rows of different trees commute, the order inside
// a tree is the solver's row order); a step with a row that couples two trees falls back to the serial sweep.
// (BIG: the build holds the solver forms that need more than 256 registers -- pgs_tall_registers, the two-position
// pgs_schedule_residual -- for models whose LDS image holds a CU to one wave per SIMD anyway, pgs_roomy.  A specialised
// kernel is built with BIG and sheds them as dead code when its model is not roomy; the generic kernels come in both
// kinds (mjrl_capi.hip), because a callee's registers count for the whole kernel: one generic kernel with the big forms
// ran EVERY model at one wave per SIMD, the 2-agent level at 13.4 M env-steps/s instead of 21.)
// `few` (StepArgs::few): a batch of at most one wave per SIMD is roomy whatever its model -- a launch of 1024 copies of
// the 2-agent level lasted 130-140 us because of ONE copy on the 32-lane solver (21 rows in a tree, 90-100 sweeps: 1 us
// per sweep) while every other wave was done after 80; with two rows per lane (pgs_tall_registers) that copy takes 95.
template <bool DIAG, bool BIG>
__device__ inline void stage_pgs(const DevModel& m, const Lay& l, const LaneK& K, const RowK& RK, real* S, int L,
                                 Stamps* stamps, bool few_arg) {
#if defined(MJRL_SPEC)
#ifdef MJRL_FEW
  constexpr bool few = true;
#else
  constexpr bool few = false;
#endif
  (void)few_arg;
#else
  const bool few = few_arg;
#endif
  const bool roomy = BIG && (few || pgs_roomy(m, l));
#define MJ_SUBSTAMP(k)                                                     \
  if constexpr (DIAG) if (stamps) {                                        \
    unsigned long long t_now = wv::clock();                                \
    if (L == (k)) stamps->mine += t_now - stamps->prev;                    \
    stamps->prev = t_now;                                                  \
  }
  int* I = (int*)(S + l.ints);
  int nefc = I[I_NEFC];
  const int mydof = RK.dof;
  const bool dof = mydof >= 0;
  // the lane's coefficient in row r with info word `info` (0 where the row does not touch the lane's dof)
  auto coef = [&](int r, int info) -> real {
    int rt = (info >> CHAIN_BITS) - 2;
    int sl = (m.rowmap && rt >= 0) ? (rt == (L >> 4) ? (L & 15) : -1) : row_slot(info, RK.below, RK.depth);
    return (dof && sl >= 0) ? S[l.J + JW * r + sl] : 0.0;
  };
  real dinv = dof ? S[l.Dinv + mydof] : 0.0;
  real u = 0;
  if (nefc == 0) {
    if (L == 0) { I[I_NITER] = 0; I[I_COST] = 1; }
    if (L < m.nv) { S[l.qfc + L] = 0; S[l.qacc + L] = S[l.qaccs + L]; }
    wv::sync();
    return;
  }
  // per-tree row lists (in the row-id array, which the finished row build no longer needs): list[base_t + rank] = row
  const int mytree = L >> 4;
  int cnt_my = 0, base_my = 0, tmax = 0;
  int cnt_w = 0, base_w = 0;        // the same for the lane's tree in the wide map of pgs_wide_registers
  bool cross = false;
  int sched_len = 0;                // sweep length of the coupled schedule (pgs_coupled_schedule)
  if (m.rowmap) {
    int total = 0;
    for (int t = 0; t < m.ntree; t++) {
      int cnt = 0;
      for (int base = 0; base < nefc; base += 64) {
        int r = base + L;
        int info = r < nefc ? I[l.i_rowinfo + r] : (1 << CHAIN_BITS);
        int rt = (info >> CHAIN_BITS) - 2;
        unsigned long long mask = wv::ballot(rt == t);
        if (rt == t) I[l.i_rowid + total + cnt + wv::popc(mask & ((1ull << L) - 1ull))] = r;
        cnt += wv::popc(mask);
        if (t == 0) cross |= wv::ballot(rt == -2) != 0ull;
      }
      if (t == mytree) { cnt_my = cnt; base_my = total; }
      if (t == ((L >> 4) & 1)) { cnt_w = cnt; base_w = total; }
      total += cnt;
      tmax = cnt > tmax ? cnt : tmax;
    }
  }
  tmax = wv::first_int(tmax);       // uniform by construction (ballot counts)
  wv::sync();
  MJ_SUBSTAMP(ST_PGS_LISTS)
  const bool wide = !roomy && m.rowmap && !cross && tmax > 16 && tmax <= 32 && m.ntree <= 2;
#ifdef MJRL_NO_TALL       // (experiments: the schedule sweep for these copies, as before round 3)
  const bool tall = false;
#else
  const bool tall = roomy && m.rowmap && !cross && tmax > 16 && tmax <= 32 && m.ntree <= 4;     // pgs_tall_registers
#endif
  const bool in_registers = (m.rowmap && !cross && tmax <= 16) || wide || tall;
  // what a sweep of this copy costs relative to one of the 16-row register solver (for the longest-first dispatch: rows x
  // sweeps of a copy on the wide, schedule or serial path stand for three to four times the wave time)
  // (a tree of 13+ rows is three contacts from the wide path: such a copy counts double, so that it is not the one the
  // launch waits for if it crosses over in the step the key predicts)
#ifndef MJRL_NEAR_WIDE
#define MJRL_NEAR_WIDE 13
#endif
  if (L == 0) I[I_COST] = (in_registers && !wide && !tall) ? (tmax >= MJRL_NEAR_WIDE ? 2 : 1) : (tall ? 2 : 4);
  if (!in_registers) {
    // warm start: keep the forces implied by last step's acceleration only if they beat f = 0
    if (dof)
      for (int r = 0; r < nefc; r++) u += coef(r, I[l.i_rowinfo + r]) * S[l.row + ROW_STRIDE * r + ROW_F];
    real part = 0.5 * dinv * u * u;
    for (int r = L; r < nefc; r += 64) {
      const real* R = S + l.row + ROW_STRIDE * r;
      part += 0.5 * R[ROW_R] * R[ROW_F] * R[ROW_F] + R[ROW_F] * R[ROW_B];
    }
    real cost = wv::sum(part);
    if (cost > 0) {
      u = 0;
      for (int r = L; r < nefc; r += 64) S[l.row + ROW_STRIDE * r + ROW_F] = 0;
    }
    wv::sync();
  }
  MJ_SUBSTAMP(ST_PGS_WARM)
  real scale = 1.0 / (m.meaninertia * (m.nv > 1 ? m.nv : 1));
  int iter = 0;
  if (wide) {
    const int wtree = (L >> 4) & 1;
    const int wadr0 = wtree < m.ntree ? m.tree_dofadr[wtree] : 0;
    const WideOut wo = pgs_wide_registers(S, I, L, cnt_w, base_w, tmax, dof, l.i_rowid, l.row, l.J, l.Dinv, m.iterations, wadr0,
                                          m.tolerance, scale, iter);
    u = wo.u; iter = wv::first_int(wo.iter);
    wv::sync();
  } else if (tall) {
    const int tadr0 = mytree < m.ntree ? m.tree_dofadr[mytree] : 0;
    const TallOut to = pgs_tall_registers(S, I, L, mytree < m.ntree ? cnt_my : 0, base_my, tmax, dof, l.i_rowid, l.row, l.J, l.Dinv,
                                          m.iterations, tadr0, m.ntree, m.tolerance, scale, iter);
    u = to.u; iter = wv::first_int(to.iter);
    wv::sync();
  } else if (in_registers) {
    // Residual form, all in registers: with at most 16 rows per tree, lane k of a tree's 16 lanes owns the tree's
    // k-th row -- its force f_k, its row of AR = B D^-1 B' + diag(R) and its residual r_k = (AR f)_k + b_k.  A
    // Gauss-Seidel step on row k is a handful of operations in lane k, one DPP broadcast of the force change and one
    // multiply-add per lane (r_j += AR_jk * delta): no reduction and no LDS access inside the sweeps.  The trees
    // step through their rows side by side.
    const int kme = L & 15;
    const bool has_row = kme < cnt_my;
    const int myrow = has_row ? I[l.i_rowid + base_my + kme] : 0;
    real* Rm = S + l.row + ROW_STRIDE * myrow;
    real fi = has_row ? Rm[ROW_F] : 0.0, bi = has_row ? Rm[ROW_B] : 0.0, Ri = has_row ? Rm[ROW_R] : 0.0;
    const real aii = has_row ? Rm[ROW_ARII] : 1.0;
    const real ainv = 1.0 / aii;
    const int adr0 = wv::bcast16i<0>(mydof);            // first dof of the lane's tree
    real W[16], A[16];
#pragma unroll
    for (int d = 0; d < 16; d++)
      W[d] = has_row ? S[l.J + JW * myrow + d] * S[l.Dinv + adr0 + d] : 0.0;     // (slots past the tree's dofs hold 0)
#pragma unroll
    for (int k = 0; k < 16; k++) A[k] = 0;
    // row k of the tree is owned by lane k of the tree's 16 lanes: its id comes over DPP, not from the list in LDS
#define MJ_ASTEP(KK)                                                                  \
      if (KK >= tmax) break;                                                          \
      {                                                                               \
        const real* Bk = S + l.J + JW * wv::bcast16i<KK>(myrow);                      \
        real p0 = 0, p1 = 0, p2 = 0, p3 = 0;      /* four partial sums: the 16 terms form chains of 4 */ \
        _Pragma("unroll")                                                             \
        for (int d = 0; d < 16; d += 4) {                                             \
          p0 += W[d] * Bk[d]; p1 += W[d + 1] * Bk[d + 1]; p2 += W[d + 2] * Bk[d + 2]; p3 += W[d + 3] * Bk[d + 3]; \
        }                                                                             \
        real acc = (p0 + p1) + (p2 + p3);                                             \
        if (KK >= cnt_my) acc = 0;                                                    \
        if (KK == kme) acc += Ri;                                                     \
        A[KK] = acc;                                                                  \
      }
    do {
      MJ_ASTEP(0) MJ_ASTEP(1) MJ_ASTEP(2) MJ_ASTEP(3) MJ_ASTEP(4) MJ_ASTEP(5) MJ_ASTEP(6) MJ_ASTEP(7)
      MJ_ASTEP(8) MJ_ASTEP(9) MJ_ASTEP(10) MJ_ASTEP(11) MJ_ASTEP(12) MJ_ASTEP(13) MJ_ASTEP(14) MJ_ASTEP(15)
    } while (0);
#undef MJ_ASTEP
    real r = bi;
#define MJ_RINIT(KK) if (KK >= tmax) break; r += A[KK] * wv::bcast16<KK>(fi);
    do {
      MJ_RINIT(0) MJ_RINIT(1) MJ_RINIT(2) MJ_RINIT(3) MJ_RINIT(4) MJ_RINIT(5) MJ_RINIT(6) MJ_RINIT(7)
      MJ_RINIT(8) MJ_RINIT(9) MJ_RINIT(10) MJ_RINIT(11) MJ_RINIT(12) MJ_RINIT(13) MJ_RINIT(14) MJ_RINIT(15)
    } while (0);
#undef MJ_RINIT
    // warm start: keep last step's forces only if they beat f = 0; cost(f) = 1/2 f'AR f + f'b = 1/2 f'(r + b)
    {
      real cost = wv::rows_sum(wv::sum16(0.5 * fi * (r + bi)), m.ntree);
      const bool cold = cost > 0;
      fi = cold ? 0.0 : fi;
      r = cold ? bi : r;
    }
    // The sweeps work on the scaled residual s = r / AR_kk and on AR's rows scaled likewise, which takes one multiply
    // out of every row step's dependent chain (f - r/AR_kk becomes f - s).
    real sr = r * ainv;
#pragma unroll
    for (int k = 0; k < 16; k++) A[k] *= ainv;
    MJ_SUBSTAMP(ST_PGS_SETUP)
    const real haii = 0.5 * aii;
    // The sweep is bound by instruction issue and by its dependent chain, so a row step carries the minimum: subtract,
    // max, subtract, broadcast, multiply-add, and in the row's own lane a capture of the new force and of the residual
    // it was computed from.  What the reference does per step besides -- the cost change 0.5 d^2 A_kk + d r_k, summed
    // into the sweep's improvement, and the guard that refuses a step which would raise the cost (change > 1e-10) --
    // is evaluated once per sweep from the captured values, and ONE SWEEP LATE: the 64-lane sum of sweep n's
    // improvement and the branch on it run in the shadow of the first two row steps of sweep n+1.  When sweep n turns
    // out to have converged, the two steps are dropped (the forces go back to the end of sweep n) -- the reference's
    // result and sweep count exactly.  Exact arithmetic never raises the cost; if rounding did in some step, the
    // solve resumes from the start of that sweep with guarded sweeps that check at once.
    real f_start = fi, s_start = sr;           // start of the sweep in progress
    real f_prev = fi, s_prev = sr;             // start of the sweep before it
    real c_prev = 0;                           // cost change of the lane's row in the sweep before (pending decision)
    bool pending = false, guarded = false;
    while (iter < m.iterations) {
      // (row steps past the last row are no-ops -- their lanes hold f = s = 0 and a zero column of AR -- so the exit
      // test, a scalar compare and a branch that costs about a third of a step, is made once per group of steps:
      // groups of two for the first four rows, of four after that; the tests are made on per-sweep copies the
      // optimiser cannot see through, or it would precompute a lane mask per step and spill them)
      const int kme_s = wv::opaque_lane(kme), tmax_s = wv::opaque_uniform(tmax);
      f_start = fi; s_start = sr;
      // A row step changes its force by d = max(f - s, 0) - f = max(-s, -f): one max on the negated residual, then
      // the broadcast and ONE fused multiply-add (an explicit fma: the build keeps contraction off everywhere else, but
      // this update sits on the dependent chain of every row step -- max, broadcast, fma instead of max, broadcast,
      // multiply, subtract -- and its residuals are the solver's own running sums, not a quantity the oracle forms) --
      // the whole dependent chain of a step.  (The reference forms d as the
      // rounded difference of the rounded new force; max(-s, -f) is that quantity without the two roundings.  The new
      // force itself, max(f - s, 0), is formed after the sweep from the residual captured in the row's step, bit for
      // bit as the reference forms it, and with it the cost change of the row.)
      real ns = -sr, nss = ns;
      const real nf = -fi;
      // The stop test of the sweep before.  Its improvement is a 64-lane sum of the rows' cost decreases, as many
      // instructions as three row steps -- but every term is >= -1e-10 (a step that would raise the cost by more is
      // refused), so ONE row whose own decrease exceeds the threshold (by more than the other rows could take back)
      // already decides "not converged", and that is what nearly every sweep of a long solve looks like: two compares
      // and a scalar test, the sum only when no single row decides.  The sum, the branch and the sweep count are the
      // reference's.
      // (the two lane masks are formed here; the branch on them comes after the first two row steps, whose work covers
      // the vector-to-scalar latency -- when the sweep before turns out to have converged those two steps are dropped)
      unsigned long long refused = 0ull, deciding = 1ull;
      if (pending) { refused = wv::ballot(c_prev > 1e-10); deciding = wv::ballot((-c_prev - 1e-8) * scale >= m.tolerance); }
#define MJ_FSTEP(KK)                                                                  \
        {                                                                             \
          real db = wv::bcast16<KK>(fmax(ns, nf));                                    \
          if (kme_s == KK) nss = ns;                                                  \
          ns = __builtin_fma(-A[KK], db, ns);                                         \
        }
      MJ_FSTEP(0) MJ_FSTEP(1)
      if (pending && (refused != 0ull || deciding == 0ull)) {
        const real improvement = wv::rows_sum(wv::sum16(-c_prev), m.ntree);
        if (refused != 0ull || wv::ballot(improvement * scale < m.tolerance)) {
          if (refused != 0ull) { fi = f_prev; sr = s_prev; iter--; guarded = true; }   // redo that sweep
          else { sr = s_start; }                                                       // it had converged
          pending = false;
          break;
        }
      }
      do {
        if (2 >= tmax_s) break;
        MJ_FSTEP(2) MJ_FSTEP(3)
        if (4 >= tmax_s) break;
        MJ_FSTEP(4) MJ_FSTEP(5) MJ_FSTEP(6) MJ_FSTEP(7)
        if (8 >= tmax_s) break;
        MJ_FSTEP(8) MJ_FSTEP(9) MJ_FSTEP(10) MJ_FSTEP(11)
        if (12 >= tmax_s) break;
        MJ_FSTEP(12) MJ_FSTEP(13) MJ_FSTEP(14) MJ_FSTEP(15)
      } while (0);
#undef MJ_FSTEP
      sr = -ns;
      const real ss = -nss;
      fi = fmax(f_start - ss, 0.0);
      const real dsweep = fi - f_start;
      c_prev = dsweep * dsweep * haii + dsweep * (ss * aii);
      f_prev = f_start; s_prev = s_start;
      pending = true;
      iter++;
    }
    if (pending && wv::ballot(c_prev > 1e-10)) {      // the sweep cap was reached and the last sweep has a refused step
      fi = f_prev; sr = s_prev; iter--; guarded = true;
    }
    while (guarded && iter < m.iterations) {
      real imp = 0;
      const int kme_g = wv::opaque_lane(kme);
#define MJ_GSTEP(KK)                                                                  \
      if (KK < tmax) {                                                                \
        real fn = fmax(fi - sr, 0.0);                                                 \
        real delta = fn - fi;                                                         \
        real change = delta * delta * haii + delta * (sr * aii);                      \
        bool act = kme_g == KK && has_row && !(change > 1e-10);                       \
        if (!act) { delta = 0; change = 0; fn = fi; }                                 \
        fi = fn;                                                                      \
        imp -= change;                                                                \
        sr += A[KK] * wv::bcast16<KK>(delta);                                         \
      }
      MJ_GSTEP(0) MJ_GSTEP(1) MJ_GSTEP(2) MJ_GSTEP(3) MJ_GSTEP(4) MJ_GSTEP(5) MJ_GSTEP(6) MJ_GSTEP(7)
      MJ_GSTEP(8) MJ_GSTEP(9) MJ_GSTEP(10) MJ_GSTEP(11) MJ_GSTEP(12) MJ_GSTEP(13) MJ_GSTEP(14) MJ_GSTEP(15)
#undef MJ_GSTEP
      iter++;
      if (wv::rows_sum(wv::sum16(imp), m.ntree) * scale < m.tolerance) break;
    }
    MJ_SUBSTAMP(ST_PGS_SWEEPS)
    if (has_row) Rm[ROW_F] = fi;
    // u = B' f for the lane's dof
    u = 0;
#define MJ_USTEP(KK)                                                                  \
      if (KK >= tmax) break;                                                          \
      {                                                                               \
        real fk = wv::bcast16<KK>(fi);                                                \
        const int rk = wv::bcast16i<KK>(myrow);                                       \
        if (dof && KK < cnt_my) u += S[l.J + JW * rk + kme] * fk;                     \
      }
    do {
      MJ_USTEP(0) MJ_USTEP(1) MJ_USTEP(2) MJ_USTEP(3) MJ_USTEP(4) MJ_USTEP(5) MJ_USTEP(6) MJ_USTEP(7)
      MJ_USTEP(8) MJ_USTEP(9) MJ_USTEP(10) MJ_USTEP(11) MJ_USTEP(12) MJ_USTEP(13) MJ_USTEP(14) MJ_USTEP(15)
    } while (0);
#undef MJ_USTEP
    wv::sync();
  } else if (m.rowmap && (cross || (roomy && tmax > 16 && tmax <= 32)) &&
             pgs_coupled_schedule(m, l, S, L, nefc, sched_len)) {
    // (also a copy without coupling rows but with 17..32 rows in a tree of a model with more than two trees, where the
    // two-positions-per-lane form below is available: its plain per-tree lists are a schedule too)
    // Some rows couple two trees (agent against agent).  The trees still sweep side by side: every tree walks its own
    // list of rows, a coupling row sits in the lists of both its trees AT THE SAME POSITION (the shorter list is padded
    // with empty entries, pgs_coupled_schedule), and at such a position the two trees add their halves of the row's
    // residual through the LDS crossbar and then take the same step.  Rows of different trees between two coupling
    // rows commute, so this is the serial order of the reference as far as any row can tell.  (The serial sweep
    // below took 10x the wave time of the register solver, and with one such copy the launch took twice as long.)
    const bool leader = (L & 15) == 0;
    const int C = m.njmax / m.ntree;
    const Tab T = make_tab(m, l, S);
    // two positions per lane only where the LDS image already holds the CU to four copies (one wave per SIMD): the
    // form needs more than 256 registers
    if (sched_len <= 16 || (roomy && sched_len <= 32)) {
      SchedArgs w;
      w.o_rowid = l.i_rowid; w.o_rowinfo = l.i_rowinfo; w.o_row = l.row; w.o_J = l.J; w.o_tab = l.tab;
      w.base = mytree < m.ntree ? mytree * C : 0; w.len = mytree < m.ntree ? sched_len : 0; w.iterations = m.iterations;
      w.tab_dtree = T.dtree; w.tab_bytes = T.bytes ? 1 : 0;
      w.depth = RK.depth; w.below = RK.below;
      w.tolerance = m.tolerance; w.scale = scale; w.dinv = dinv; w.u = u;
      w.o_Dinv = l.Dinv; w.adr0 = mytree < m.ntree ? m.tree_dofadr[mytree] : 0;
      int it = iter;        // (a local of its own: the counter of the other solver paths never has its address taken)
      if (!BIG || sched_len <= 16) u = pgs_schedule_residual<16>(S, I, L, dof, w, m.ntree, it);
      else u = pgs_schedule_residual32(S, I, L, dof, w, m.ntree, it, &it);
      iter = it;
      wv::sync();
    } else {
    struct Rec { int i, info; bool has; real bid, fi, Ri, bi, aii, ainv; };
    auto fetch = [&](int sidx) {
      Rec r;
      const int e = mytree < m.ntree ? I[l.i_rowid + mytree * C + sidx] : -1;
      r.has = e >= 0;
      r.i = r.has ? e : 0;
      r.info = I[l.i_rowinfo + r.i];
      const real* R = S + l.row + ROW_STRIDE * r.i;
      r.bid = r.has ? coef(r.i, r.info) : 0.0;
      r.fi = R[ROW_F]; r.Ri = R[ROW_R]; r.bi = R[ROW_B]; r.aii = R[ROW_ARII];
      r.ainv = 1.0 / r.aii;
      return r;
    };
    Rec nxt = fetch(0);
    while (iter < m.iterations) {
      real imp = 0;
      for (int sidx = 0; sidx < sched_len; sidx++) {
        Rec c = nxt;
        nxt = fetch(sidx + 1 < sched_len ? sidx + 1 : 0);
        real res = wv::sum16(c.bid * dinv * u);
        const bool coupling = c.has && (c.info >> CHAIN_BITS) == 0;
        bool counted = true;
        if (wv::ballot(coupling)) {
          // the other tree of the row holds the other half of the sum (a + b == b + a: both trees get the same bits)
          int partner = L;
          if (coupling) {
            const int t1 = T.dof_tree(c.info & 63), t2 = T.dof_tree(((c.info >> 9) & 127) - 1);
            partner = 16 * (t1 == mytree ? t2 : t1) + (L & 15);
            counted = mytree == (t1 < t2 ? t1 : t2);          // its cost change enters the sweep's improvement once
          }
          const real other = wv::shfl(res, partner);
          if (coupling) res += other;
        }
        res = res + c.Ri * c.fi + c.bi;
        real fn = c.fi - res * c.ainv;
        if (fn < 0) fn = 0;
        real delta = fn - c.fi;
        real change = 0.5 * delta * delta * c.aii + delta * res;
        if (change > 1e-10 || !c.has) { fn = c.fi; delta = 0; change = 0; }
        if (counted) imp -= change;
        u += delta * c.bid;
        if (leader && c.has) S[l.row + ROW_STRIDE * c.i + ROW_F] = fn;
        if (nxt.has && c.has && nxt.i == c.i) nxt.fi = fn;      // the same row comes again (a single-entry list)
      }
      iter++;
      real improvement = wv::rows4_sum(imp);
      if (improvement * scale < m.tolerance) break;
    }
    wv::sync();
    }
  } else if (m.rowmap && !cross) {
    const bool leader = (L & 15) == 0;
    // software pipeline: the record of the tree's next row is fetched while the current row is processed
    // (the row list is the same in every sweep; only the force F changes, and a row's F is rewritten by this
    // tree's lanes alone)
    struct Rec { int i; bool has; real bid, fi, Ri, bi, aii, ainv; };
    auto fetch = [&](int sidx) {
      Rec r;
      r.has = sidx < cnt_my;
      r.i = r.has ? I[l.i_rowid + base_my + sidx] : 0;
      const real* R = S + l.row + ROW_STRIDE * r.i;
      r.bid = (dof && r.has) ? S[l.J + JW * r.i + (L & 15)] : 0.0;      // tree-local rows: the lane's own slot
      r.fi = R[ROW_F]; r.Ri = R[ROW_R]; r.bi = R[ROW_B]; r.aii = R[ROW_ARII];
      r.ainv = 1.0 / r.aii;       // one row ahead of its use, off the sweep's dependent chain
      return r;
    };
    Rec nxt = fetch(0);
    while (iter < m.iterations) {
      real imp = 0;
      for (int sidx = 0; sidx < tmax; sidx++) {
        Rec c = nxt;
        nxt = fetch(sidx + 1 < tmax ? sidx + 1 : 0);
        real res = wv::sum16(c.bid * dinv * u) + c.Ri * c.fi + c.bi;
        real fn = c.fi - res * c.ainv;
        if (fn < 0) fn = 0;
        real delta = fn - c.fi;
        real change = 0.5 * delta * delta * c.aii + delta * res;
        if (change > 1e-10 || !c.has) { fn = c.fi; delta = 0; change = 0; }
        imp -= change;
        u += delta * c.bid;
        if (leader && c.has) S[l.row + ROW_STRIDE * c.i + ROW_F] = fn;
        if (nxt.i == c.i) nxt.fi = fn;      // the same row comes again (a tree with a single row)
      }
      iter++;
      real improvement = wv::rows4_sum(imp);
      if (improvement * scale < m.tolerance) break;
    }
    wv::sync();
  } else {
    const int width = m.rowmap ? 64 : (m.nv <= 16 ? 16 : (m.nv <= 32 ? 32 : 64));
    while (iter < m.iterations) {
      real improvement = 0;
      for (int i = 0; i < nefc; i++) {
        const real* R = S + l.row + ROW_STRIDE * i;
        real bid = coef(i, I[l.i_rowinfo + i]);
        real fi = R[ROW_F], Ri = R[ROW_R], bi = R[ROW_B], aii = R[ROW_ARII], ainv = 1.0 / aii;
        // reduce over the lanes that hold dofs, then hand lane 0's sum to the whole wave
        real res = wv::first(wv::sum_n(bid * dinv * u, width)) + Ri * fi + bi;
        real fn = fi - res * ainv;
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
  }
  if (L == 0) I[I_NITER] = iter;
  MJ_SUBSTAMP(ST_PGS_SWEEPS)
#undef MJ_SUBSTAMP
  // back to joint space: qfrc_constraint = L' u ; qacc = qacc_smooth + L^-1 D^-1 u
  if (m.rowmap) {
    // (L' u)_d = u_d + sum over descendants k of L[k][d] u_k: the backward-solve pattern with the ORIGINAL u
    real q = u;
#define MJ_QSTEP(KK) if (KK < m.maxtreedof) { real uk = wv::bcast16<KK>(u); q += S[RK.eb[KK]] * uk; }
    MJ_QSTEP(1) MJ_QSTEP(2) MJ_QSTEP(3) MJ_QSTEP(4) MJ_QSTEP(5) MJ_QSTEP(6) MJ_QSTEP(7) MJ_QSTEP(8)
    MJ_QSTEP(9) MJ_QSTEP(10) MJ_QSTEP(11) MJ_QSTEP(12) MJ_QSTEP(13) MJ_QSTEP(14) MJ_QSTEP(15)
#undef MJ_QSTEP
    if (dof) S[l.qfc + mydof] = q;
  } else {
    if (dof) S[l.x + mydof] = u;
    wv::sync();
    if (L < m.nv) {
      real q = S[l.x + L];
      for (int c = 0; c < K.d_descnum; c++) q += S[l.LD + m.desc_Madr[K.d_descadr + c]] * S[l.x + m.desc_row[K.d_descadr + c]];
      S[l.qfc + L] = q;
    }
  }
  if (m.rowmap) {
    real x = solve_rows(m, RK, S, l.LD, l.Dinv, u, false, true, true);
    if (dof) {
      real a = S[l.qaccs + mydof] + x;
      S[l.qacc + mydof] = a;     // also next step's warm start (same slots)
    }
    wv::sync();
  } else {
    wv::sync();
    solve_ld(m, K, S, l.LD, l.Dinv, l.x, L, false, true, true);
    if (L < m.nv) {
      real a = S[l.qaccs + L] + S[l.x + L];
      S[l.qacc + L] = a;
    }
    wv::sync();
  }
}

// ------------------------------------------------------------------ sensors (mj_forward's sensor stage)
__device__ inline void stage_sensors(const DevModel& m, const Lay& l, const LaneK& K, const SensK& SK0, real* S, int L) {
  int* I = (int*)(S + l.ints);
  const Tab T = make_tab(m, l, S);
  if (m.has_accel) stage_velocity(m, l, K, S, L, true);
  // Sensor s's record lives in lane s (64 sensors per pass): the chain sensor -> site -> body was followed for all
  // sensors at once on the host (lane records, SensK; a model with more than 64 sensors reads the later chunks' here),
  // and the loop over the sensors reads lane s with v_readlane.
  for (int s0 = 0; s0 < m.nsensor; s0 += 64) {
    SensK SK = SK0;
    if (s0 > 0) load_sensor_constants(m, s0, L, SK);
    const int k_adr = SK.adr, k_type = SK.type;
    const real k_cut = SK.cut;
    // the lane's geom as a ray target, once for all rangefinders
    const real rg_alpha = SK.g_alpha;
    const int rg_body0 = SK.g_body, rg_type0 = SK.g_type;
    const real rg_rb0 = SK.g_rb;
    const V3 rg_size0 = SK.g_size, rg_gpos = SK.g_pos;
    const Quat rg_gquat = SK.g_quat;
    const int k_body = SK.body;
    const V3 k_pos = SK.pos;
    const Quat k_quat = SK.quat;
    const real k_size = SK.size;
    const bool any_ray = wv::ballot(s0 + L < m.nsensor && k_type == SENS_RANGEFINDER) != 0ull;
    int rg_body = -1, rg_type = -1;
    real rg_rb = 0;
    V3 rg_pos = v3(0, 0, 0), rg_size = v3(0, 0, 0);
    M3 rg_mat = qmat(ldq(S + l.xquat));
    if (any_ray && L < m.ngeom && rg_alpha != 0) {
      rg_body = rg_body0; rg_type = rg_type0; rg_rb = rg_rb0;
      rg_size = rg_size0;
      // (the arithmetic of geom_frame())
      Quat bq = ldq(S + l.xquat + 4 * rg_body0);
      rg_pos = ld3(S + l.xpos + 3 * rg_body0) + rot(bq, rg_gpos);
      rg_mat = qmat(qmul(bq, rg_gquat));
    }
    // the site's world frame, for every sensor at once in the sensor's lane (it is the same for all lanes of the wave:
    // computed inside the loop below it cost every sensor a hundred instructions of the whole wave)
    const Quat k_bq = ldq(S + l.xquat + 4 * k_body);
    const V3 k_sp = ld3(S + l.xpos + 3 * k_body) + rot(k_bq, k_pos);
    const M3 k_sm = qmat(qmul(k_bq, k_quat));
    const int ns = m.nsensor - s0 < 64 ? m.nsensor - s0 : 64;
    for (int j = 0; j < ns; j++) {
      const int adr = wv::lane_int(k_adr, j), body = wv::lane_int(k_body, j), type = wv::lane_int(k_type, j);
      const real cutoff = wv::lane_value(k_cut, j);
      const V3 sp = v3(wv::lane_value(k_sp.x, j), wv::lane_value(k_sp.y, j), wv::lane_value(k_sp.z, j));
      if (type == SENS_RANGEFINDER) {
        const V3 vec = v3(wv::lane_value(k_sm.m[2], j), wv::lane_value(k_sm.m[5], j), wv::lane_value(k_sm.m[8], j));   // col(sm, 2)
        real best = 1e300;
        if (rg_type >= 0 && rg_body != body) {
          // a geom whose bounding sphere the ray misses (or that lies wholly behind the ray's origin) cannot be hit: it
          // takes no type-specific test, and a type with no candidate left in the wave costs nothing
          int gt = rg_type;
          if (gt != GEOM_PLANE) {
            V3 rel = rg_pos - sp;
            real t = dot(rel, vec), d2 = dot(rel, rel) - t * t;
            if (d2 > rg_rb * rg_rb * (1.0 + 1e-9) + 1e-12 || t + rg_rb < -1e-9) gt = -1;
          }
#ifdef MJRL_EXP_NORAY
          real x = gt >= 0 ? rg_pos.x : -1.0;
#else
          real x = ray_geom(gt, rg_pos, rg_mat, rg_size, sp, vec);
#endif
          if (x >= 0) best = x;
        }
        for (int g = 64 + L; g < m.ngeom; g += 64) {      // (targets past the 64th geom: the same tests, records from the model)
          if (m.geom_rgba[4 * g + 3] == 0 || m.geom_bodyid[g] == body) continue;
          V3 gp; Quat gq;
          geom_frame(m, l, S, g, gp, gq);
          int gt = m.geom_type[g];
          const real rb = m.geom_rbound[g];
          if (gt != GEOM_PLANE) {
            V3 rel = gp - sp;
            real t = dot(rel, vec), d2 = dot(rel, rel) - t * t;
            if (d2 > rb * rb * (1.0 + 1e-9) + 1e-12 || t + rb < -1e-9) gt = -1;
          }
          real x = ray_geom(gt, gp, qmat(gq), ld3(m.geom_size + 3 * g), sp, vec);
          if (x >= 0 && x < best) best = x;
        }
        best = wv::min_pos(best);
        real out = best > 1e299 ? -1.0 : best;
        if (cutoff > 0 && out > cutoff) out = cutoff;
        if (L == 0) S[l.sens + adr] = out;
      } else if (type == SENS_TOUCH) {
        const real site_size = wv::lane_value(k_size, j);
        real part = 0;
        int ncon = I[I_NCON];
        if (L < ncon) {
          int c = L, a = I[l.i_conadr + c];
          int g1 = I[l.i_cong1 + c], g2 = I[l.i_cong2 + c];
          int b1 = T.geom_body(g1), b2 = T.geom_body(g2);          // (LDS structure tables)
          if (a >= 0 && (b1 == body || b2 == body)) {
            int dim = T.geom_condim(g1) > T.geom_condim(g2) ? T.geom_condim(g1) : T.geom_condim(g2);
            int rows = dim == 1 ? 1 : 2 * ((dim < 3 ? dim : 3) - 1);
            real fn = 0;
            for (int r = 0; r < rows; r++) fn += S[l.row + ROW_STRIDE * (a + r) + ROW_F];
            const real* C = S + l.con + CON_STRIDE * c;
            V3 ray = ld3(C + CON_FRAME) * (b2 == body ? -1.0 : 1.0);
            if (fn > 0 && ray_sphere_at(sp, site_size, ld3(C + CON_POS), ray) >= 0) part = fn;
          }
        }
        real out = wv::sum(part);
        if (cutoff > 0 && out > cutoff) out = cutoff;
        if (L == 0) S[l.sens + adr] = out;
      } else if (type == SENS_ACCELEROMETER) {
        M3 sm;
#pragma unroll
        for (int e = 0; e < 9; e++) sm.m[e] = wv::lane_value(k_sm.m[e], j);
        if (L == 0) {
          int t = T.body_tree(body);
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
        // (a select, not col(sm, c): a run-time column index would put the matrix in scratch memory)
        const V3 own = c == 0 ? col(k_sm, 0) : (c == 1 ? col(k_sm, 1) : col(k_sm, 2));     // (the sensor's lane holds its axis)
        const V3 axis = v3(wv::lane_value(own.x, j), wv::lane_value(own.y, j), wv::lane_value(own.z, j));
        if (L == 0) st3(S + l.sens + adr, axis);
      }
    }
  }
  wv::sync();
}

// ------------------------------------------------------------------ integrator
// What the integrator and the observation gather read from HBM / the model, fetched before the sensor stage: the first
// schedule words of the second factorisation, the lane's joint record, and the first two gather codes of the lane.
struct EulerK {
  FactorRing ring;
  int qa, da, jtype;
  int gcode[2];
};
__device__ __forceinline__ void load_euler_constants(const DevModel& m, const StepArgs& a, int L, EulerK& E) {
  // (fetched whether or not this launch integrates: filled in one branch and zeroed in the other, the ring became an
  // object on the stack addressed through a per-branch offset)
  RecR RR;
  fetch_lane_record(a.lane_rec, REC_R, L, RR);
  E.ring = RR.ring;
  RecE RE;
  fetch_lane_record(a.lane_rec, REC_E, L, RE);
  E.qa = RE.EJ.qa; E.da = RE.EJ.da; E.jtype = RE.EJ.jtype;
  // (the codes of the rows this launch writes: every agent's, or one agent's -- StepArgs::io_agent)
  const int nobs = a.io_agent1 == 0 ? a.n_agent * a.obs_dim : a.obs_dim, g0 = a.io_agent1 == 0 ? 0 : (a.io_agent1 - 1) * a.obs_dim;
#pragma unroll
  for (int u = 0; u < 2; u++) E.gcode[u] = (a.obs && 64 * u < nobs) ? a.gather[g0 + (64 * u + L < nobs ? 64 * u + L : 0)] : -1;
}

// (the caller has put the unfactorised inertia matrix back at S[l.LD..]: mkeep_put)
__device__ inline void stage_euler(const DevModel& m, const Lay& l, const LaneK& K, const RowK& RK, const EulerK& E, real* S,
                                   int L) {
  real h = m.timestep;
  bool damped = wv::ballot(L < m.nv && K.d_damping > 0) != 0ull;
  if (damped) {
    // (M + h*diag(damping)) qacc = qfrc_smooth + qfrc_constraint
    wv::sync();
    if (L < m.nv) {
      S[l.LD + K.d_Madr] += h * K.d_damping;
      S[l.x + L] = S[l.smooth + L] + S[l.qfc + L];
    }
    wv::sync();
    factor_ld(m, S, l.LD, l.Dinv, L, E.ring);
    if (m.rowmap) {
      real x = solve_rows(m, RK, S, l.LD, l.Dinv, RK.dof >= 0 ? S[l.x + RK.dof] : 0.0, true, true, true);
      wv::sync();
      if (RK.dof >= 0) S[l.x + RK.dof] = x;
      wv::sync();
    } else {
      solve_ld(m, K, S, l.LD, l.Dinv, l.x, L, true, true, true);
    }
  } else {
    if (L < m.nv) S[l.x + L] = S[l.qacc + L];
    wv::sync();
  }
  if (L < m.nv) S[l.qvel + L] += h * S[l.x + L];
  wv::sync();
  if (L < m.njnt) {
    int qa = E.qa, da = E.da;
    if (E.jtype == JNT_FREE) {
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

// One pass of the classical 4th-order Runge-Kutta step (the oracle's ora_step_rk4, operation for operation): the forward
// pass of this launch has left qacc at the trial state in LDS; fold (qvel, qacc) into the weighted sums and set up the next
// trial state X0 + h c F, or after the fourth pass the step's result X0 + h sum b_i F_i.  Positions move on the
// configuration manifold like in the Euler update; joint damping stays explicit.
__device__ inline void stage_rk4(const DevModel& m, const Lay& l, real* S, int L, int stage, real* R) {
  const real h = m.timestep;
  const real bw = (stage == 0 || stage == 3) ? 1.0 / 6.0 : 1.0 / 3.0, c = stage == 2 ? 1.0 : 0.5;
  real* x0q = R;
  real* x0v = R + m.nq;
  real* accv = x0v + m.nv;
  real* acca = accv + m.nv;
  if (stage == 0) for (int i = L; i < m.nq; i += 64) x0q[i] = S[l.qpos + i];
  if (L < m.nv) {
    const real qv = S[l.qvel + L], qa = S[l.qacc + L];
    const real v0 = stage == 0 ? qv : x0v[L];
    real sv, sa;
    if (stage == 0) { sv = bw * qv; sa = bw * qa; x0v[L] = qv; }
    else { sv = accv[L] + bw * qv; sa = acca[L] + bw * qa; }
    if (stage < 3) { accv[L] = sv; acca[L] = sa; }
    S[l.x + L] = stage < 3 ? c * qv : sv;                       // the velocity the positions move with
    S[l.qvel + L] = stage < 3 ? v0 + h * (c * qa) : v0 + h * sa;
  }
  if (stage > 0) for (int i = L; i < m.nq; i += 64) S[l.qpos + i] = x0q[i];
  wv::sync();
  if (L < m.njnt) {
    int qa = m.jnt_qposadr[L], da = m.jnt_dofadr[L];
    if (m.jnt_type[L] == JNT_FREE) {
      for (int k = 0; k < 3; k++) S[l.qpos + qa + k] += h * S[l.x + da + k];
      real len;
      V3 w = normalized(ld3(S + l.x + da + 3), &len);
      Quat q = qmul(qnormalized(ldq(S + l.qpos + qa + 3)), axis_angle(w, h * len));
      stq(S + l.qpos + qa + 3, q);
    } else {
      S[l.qpos + qa] += h * S[l.x + da];
    }
  }
  wv::sync();
}

// Longest-first dispatch, workgroup id -> copy (see StepArgs::lpt_*): walk the bucket counts from the heaviest bucket
// down (they come in with one load, lane b holding bucket b's: my_count), then take the set bit of that rank in the
// bucket's row.  Lane L counts the bits of words 2L and 2L + 1 of a block of 128 words; the prefix over the lanes comes
// from seven ballots (a lane's count is below 128), the lane that holds the bit and the bit's place in its word are
// found in scalar registers.  Returns the copy, -1 when the counts do not cover the workgroup (they sum to n_env by
// construction: such a workgroup has no copy to step), -2 when the rows disagree with the counts.
__device__ inline int lpt_copy_of(const StepArgs& a, int L, int wg, int my_count, int& bucket_out) {
  int rest = wg, bucket = -1;
  for (int b = LPT_BUCKETS - 1; b >= 0; b--) {
    int c = wv::lane_int(my_count, b);
    if (bucket < 0) {
      if (rest < c) bucket = b; else rest -= c;
    }
  }
  if (bucket < 0) return -1;
  bucket_out = bucket;
  const unsigned* row = a.lpt_mask_in + (size_t)bucket * a.lpt_words;
  const unsigned long long upto = (2ull << L) - 1ull;              // lanes 0..L
  int found = -1, before = 0;
  for (int w0 = 0; w0 < a.lpt_words && found < 0; w0 += 128) {
    const int i0 = w0 + 2 * L, i1 = i0 + 1;
    const unsigned m0 = row[i0 < a.lpt_words ? i0 : 0], m1 = row[i1 < a.lpt_words ? i1 : 0];
    const unsigned u0 = i0 < a.lpt_words ? m0 : 0u, u1 = i1 < a.lpt_words ? m1 : 0u;
    const int c = wv::popc32(u0) + wv::popc32(u1);
    int incl = 0;
    for (int bit = 0; bit < 7; bit++) incl += wv::popc(wv::ballot(((c >> bit) & 1) != 0) & upto) << bit;
    const int total = wv::lane_int(incl, 63), local = rest - before;
    if (local < total) {
      const int sel = wv::first_set(wv::ballot(incl > local));
      int r = local - (wv::lane_int(incl, sel) - wv::lane_int(c, sel));
      const unsigned s0 = (unsigned)wv::lane_int((int)u0, sel), s1 = (unsigned)wv::lane_int((int)u1, sel);
      unsigned word = s0;
      int wi = w0 + 2 * sel;
      if (r >= wv::popc32(s0)) { r -= wv::popc32(s0); word = s1; wi++; }
      int pos = 0;
      for (int sh = 16; sh >= 1; sh >>= 1) {
        const unsigned lo = word & ((1u << sh) - 1u);
        const int cc = wv::popc32(lo);
        if (r >= cc) { r -= cc; word >>= sh; pos += sh; } else word = lo;
      }
      found = 32 * wi + pos;
    }
    before += total;
  }
  return (found < 0 || found >= a.n_env) ? -2 : found;
}

// ------------------------------------------------------------------ one env copy, one step() call
// DIAG = false is the production build: the stage clock, the wave timeline, the LDS dump and the stage cuts (StepArgs::
// stamps / timeline / dbg / stop_after) do not exist in it -- as run-time branches they cost the shipped kernel a stack
// object (the clock), a hundred scalar registers' worth of spills and an exec-masked block per stage.  Launches that ask
// for any of them go to the DIAG = true build (mjrl_step_kernel_diag; a specialised kernel built with -DMJRL_DIAG).
// (inlined by force: past some size the inliner leaves it a function of its own, and the kernel then hands it the model
// -- every size a constant of a specialised build -- as a 1.3 KB struct on the stack)
template <bool DIAG, bool BIG = true>
__device__ __forceinline__ void env_step_t(const DevModel& m, const StepArgs& a_in, real* S) {
  const StepArgs& a = a_in;
  const int L = wv::lane();
  int env = wv::env_index();
  unsigned long long t_begin = 0ull;
  if constexpr (DIAG) t_begin = a.timeline ? wv::realtime() : 0ull;
  // The prologue is a chain of dependent round trips to L2 / HBM (bucket counts -> the copy's id -> the copy's rows),
  // and nothing else of the wave can start before it ends.  So every load that does not need the copy's id is issued
  // before the id is known, and every load that needs it is issued before the first of them is waited for: three round
  // trips in all (they were eleven when each block of state was fetched and stored in turn).
#ifndef MJRL_STAGE_CUT
  if constexpr (DIAG) if (a.stop_after) return;     // (a truncated launch on a kernel without the cuts must not run a whole step)
#endif
  int my_count = 0;
  if (a.lpt_count_in) my_count = a.lpt_count_in[L & (LPT_BUCKETS - 1)];
  Lay l;
  make_layout(m, l);
  // the lane's own records of the model (lane records: one group of 16-byte loads instead of ninety table reads)
  RecA RA;
  fetch_lane_record(a.lane_rec, REC_A, L, RA);
  const LaneK& K = RA.K;
  RowK RK = RA.RK;
  const TabRegs& TR = RA.TR;
  const KinK& KK = RA.KK;
  const ComK& CK = RA.CK;
  // (the dispatch tables of the launch after next are cleared here, before a workgroup without a copy can leave)
  if (a.lpt_count_clear && wv::env_index() == 0 && L < LPT_BUCKETS) a.lpt_count_clear[L] = 0;
  if (a.lpt_mask_clear && wv::env_index() < a.lpt_words && L < LPT_BUCKETS)
    a.lpt_mask_clear[(size_t)L * a.lpt_words + wv::env_index()] = 0u;
  if (a.lpt_count_in) {
    int bucket = 0;
    env = lpt_copy_of(a, L, env, my_count, bucket);
    if (env == -2 && a.overflow && L == 0) wv::atomic_add(a.overflow + 2, 1ull);
    if (env < 0) return;
    // The launch ends with its slowest copy, and a copy with hundreds of solver row steps is one long dependent
    // chain: its wave gets issue priority over the waves that share its SIMD (they fill the gaps it leaves).
    wv::set_priority(bucket >= 10 ? 3 : (bucket == 9 ? 2 : (bucket == 8 ? 1 : 0)));
  }
  // diagnostic stage clock: lane k accumulates the cycles of stage k in a register and adds them to the batch totals
  // when the wave is done -- nothing of the measurement touches memory while a stage is being timed
  Stamps clock_state;
  clock_state.prev = 0ull;
  clock_state.mine = 0;
  Stamps* stamps = nullptr;
  if constexpr (DIAG) {
    clock_state.prev = a.stamps ? wv::clock() : 0ull;
    stamps = a.stamps ? &clock_state : nullptr;
  }
  // (the stage cuts of mjrl_step_truncated exist in diagnostic builds only, -DMJRL_STAGE_CUT through MJRL_SPEC_FLAGS:
  // an exit after every stage costs the 4-agent kernel 1.4 KB of scratch per lane)
#ifdef MJRL_STAGE_CUT
#define MJ_CUT(k) if constexpr (DIAG) if (a.stop_after == (k) + 1) return;
#else
#define MJ_CUT(k)
#endif
#define MJ_STAMP_ONLY(k)                                                   \
  if constexpr (DIAG) if (stamps) {                                        \
    unsigned long long t_now = wv::clock();                                \
    if (L == (k)) stamps->mine += t_now - stamps->prev;                    \
    stamps->prev = t_now;                                                  \
  }
#define MJ_STAMP(k) MJ_STAMP_ONLY(k) MJ_CUT(k)
#define MJ_FOR(i, n) for (int i = L; i < (n); i += 64)
  // state in: the copy's rows (one element per lane and array, a second one of qpos when nq > 64), fetched whether or
  // not the copy is flagged for an in-launch reset -- the flag arrives with them
  const bool may_reset = a.reset_mask != nullptr && !a.forward_only;
  const bool may_auto = a.auto_mask != nullptr && !a.forward_only;
  int r_mask = may_reset ? (int)a.reset_mask[env] : 0;
  const int r_auto = may_auto ? (int)a.auto_mask[env] : 0;
  const int r_ts = a.forward_only ? 0 : a.timestep[env];
  const int r_ep = (a.episode && a.n_op > 0) ? a.episode[env] : 0;
  const real r_qpos0 = a.qpos[(size_t)env * m.nq + (L < m.nq ? L : 0)];
  const real r_qpos1 = m.nq > 64 ? a.qpos[(size_t)env * m.nq + (L + 64 < m.nq ? L + 64 : 0)] : 0.0;
  const real r_qvel = a.qvel[(size_t)env * m.nv + (L < m.nv ? L : 0)];
  const real r_warm = a.warm[(size_t)env * m.nv + (L < m.nv ? L : 0)];
  const real r_ctrl = m.nu > 0 ? a.ctrl[(size_t)env * m.nu + (L < m.nu ? L : 0)] : 0.0;
  // what the end of the step reads from HBM is fetched now as well, one element per lane, so that no load latency is
  // left exposed after the integrator: the step counter (above) and, for the fused plugin ops, the copy's action row
  // and its data-store row (staged in the dead bias-force vector once the integrator is done; a program too large for
  // that reads HBM directly)
  // staging area at the end of the step: the four nv-vectors from the bias forces on, all dead after the integrator
  //   [action row | data-store row | prog_f (4 per op) | prog_i (8 ints per op) | obs_len, agent_body (ints)]
  const int n_act_row = a.n_agent * a.act_dim, n_store_row = a.n_agent * a.n_slot;
  const int t_store = n_act_row, t_pf = t_store + n_store_row, t_pi = t_pf + 4 * a.n_op, t_ag = t_pi + 4 * a.n_op;
  const bool ops_staged = a.n_op > 0 && t_ag + a.n_agent <= 4 * m.nv && n_act_row <= 64 && 8 * a.n_op <= 64;
  real act_reg = 0, store_reg = 0, pf_reg = 0;
  int pi_reg = 0, len_reg = 0, body_reg = 0;
  // the action row and its scatter indices, one element per lane, issued with the state loads (the scatter below would
  // otherwise wait for two dependent loads of its own)
  const bool act_in_lanes = a.actions != nullptr && n_act_row <= 64;
  int sc_reg = -1;
  if (act_in_lanes && n_act_row > 0) {
    const int i = L < n_act_row ? L : 0;
    if (a.io_agent1 == 0) {
      act_reg = a.actions[(size_t)env * n_act_row + i];
    } else {                                      // (one agent's row in the buffer: the other agents act with 0)
      const int j = i - (a.io_agent1 - 1) * a.act_dim;
      const bool mine = j >= 0 && j < a.act_dim;
      const real v = a.actions[(size_t)env * a.act_dim + (mine ? j : 0)];
      act_reg = mine ? v : 0.0;
    }
    if (a.scatter) sc_reg = a.scatter[i];         // (lanes past the row hold entry 0: masked where it is used)
  }
  if (ops_staged) {
    if (a.store && n_store_row > 0) store_reg = a.store[(size_t)env * n_store_row + (L < n_store_row ? L : 0)];
    pf_reg = a.prog_f[L < 4 * a.n_op ? L : 0];
    pi_reg = a.prog_i[L < 8 * a.n_op ? L : 0];
    if (a.n_agent > 0) { len_reg = a.agent_obs_len[L < a.n_agent ? L : 0]; body_reg = a.agent_body[L < a.n_agent ? L : 0]; }
  }
  // -- every load of the prologue is in flight; from here on their values are used --
  if (may_auto && r_mask == 0 && r_auto != 0) r_mask = a.auto_mode == 1 ? 2 : 1;      // (an explicit flag wins)
  const int reset_kind = (may_reset || may_auto) ? wv::first_int(r_mask) : 0;
  const bool resetting = reset_kind != 0;
  const bool reset_only = reset_kind == 2;       // to the reset image, no physics frame (see StepArgs::reset_mask)
  const int ts = (a.forward_only || resetting) ? 0 : wv::first_int(r_ts);
  // the episode this step belongs to: an in-launch reset starts the next one
  const unsigned long long ep_key = (unsigned long long)(unsigned)(wv::first_int(r_ep) + (resetting ? 1 : 0)) << 32;
  if (resetting) store_reg = __builtin_nan("");          // (reset: an empty store)
  if (resetting && a.first_frame) {
    MJ_FOR(i, m.nq) S[l.qpos + i] = m.qpos0[i];
    MJ_FOR(i, m.nv) { S[l.qvel + i] = 0; S[l.warm + i] = a.reset_warm[i]; }
    MJ_FOR(i, m.nu) S[l.ctrl + i] = 0;
  } else {
    if (L < m.nq) S[l.qpos + L] = r_qpos0;
    if (m.nq > 64 && L + 64 < m.nq) S[l.qpos + L + 64] = r_qpos1;
    if (L < m.nv) { S[l.qvel + L] = r_qvel; S[l.warm + L] = r_warm; }
    if (L < m.nu) S[l.ctrl + L] = r_ctrl;
  }
  if (reset_only) MJ_FOR(i, m.nsensordata) S[l.sens + i] = a.reset_sens[i];     // (every launch of such a step: no frame computes them)
  stage_constants(m, l, S, L, TR);
  wv::sync();
  // scatter the physical part of every agent's action (mujoco_parent.py:323-332)
  if (a.actions && a.scatter && !reset_only) {
    if (act_in_lanes) {
      if (L < n_act_row && sc_reg >= 0) { if (a.scatter_mode == 0) S[l.ctrl + sc_reg] = act_reg; else S[l.qvel + sc_reg] = act_reg; }
    } else {
      MJ_FOR(it, a.n_agent * a.act_dim) {
        int idx = a.scatter[it];
        if (idx >= 0) {
          real v = a.actions[(size_t)env * a.n_agent * a.act_dim + it];
          if (a.scatter_mode == 0) S[l.ctrl + idx] = v; else S[l.qvel + idx] = v;
        }
      }
    }
    wv::sync();
  }
  MJ_STAMP(ST_LOAD)
  // One physics frame per launch (the host issues skip_frames launches, mjrl_capi.hip launch_step).  A frame loop
  // in here would make everything a frame computes from the lane id, the lane constants and the model loop-invariant:
  // hoisted, those hundreds of masks and addresses stay live for the whole kernel and spill (340 VGPRs with AGPR
  // spill space against 218 without the loop).  Between launches the state round-trips through HBM, 2.5 KB per copy.
  EulerK EK;
  EK.gcode[0] = EK.gcode[1] = -3;          // (-3: not fetched -- a launch without a physics frame reads the table below)
  // Every stage gets the lane id as a value the optimiser has not seen before (MJ_L: no instruction).  The lane masks
  // a stage derives from it (L < nv, L < nbody, L == 0 ...) are then formed where the stage uses them -- one compare --
  // instead of once at the top of the kernel for all stages: kept that long they do not fit the scalar registers, and
  // every use fetched its mask back from a spilled register with two v_readlane.
#define MJ_L wv::opaque_lane(L)
  if (a.skip_frames && !reset_only) {
    stage_kinematics(m, l, K, KK, S, MJ_L);
    MJ_STAMP(ST_KIN)
    stage_com_inertia(m, l, K, KK, CK, S, MJ_L);
    MJ_STAMP(ST_COM)
    // (model constants of a stage are fetched a stage or two ahead of it: with two waves on a SIMD nothing else hides a
    // round trip to L2 at the head of a stage)
    RecG RG;
    fetch_lane_record(a.lane_rec, REC_G, MJ_L, RG);
    const GeomK& GK = RG.GK;
    RecR RR;
    fetch_lane_record(a.lane_rec, REC_R, MJ_L, RR);
    const FactorRing& ring = RR.ring;
    MKeep Mk;
    stage_crb(m, l, K, S, MJ_L);
    mkeep_take(m, l, S, MJ_L, Mk);
    MJ_STAMP(ST_CRB)
    factor_ld(m, S, l.LD, l.Dinv, MJ_L, ring);
    MJ_STAMP(ST_FACTOR)
    stage_geoms(m, l, GK, S, MJ_L);
    MJ_STAMP(ST_GEOM)
    RecC RC;
    fetch_lane_record(a.lane_rec, REC_C, MJ_L, RC);
    const ActK& AK = RC.AK;
    stage_collision(m, l, GK, S, MJ_L);
    MJ_STAMP(ST_COLLIDE)
    stage_velocity(m, l, K, S, MJ_L, false);
    MJ_STAMP(ST_VEL)
    RK.dof = wv::opaque_lane(RK.dof);      // (the same for the mask "this lane holds a dof" of the tree-row lane map)
    stage_smooth(m, l, K, RK, AK, S, MJ_L);
    MJ_STAMP(ST_SMOOTH)
    // (a raw-row debug dump keeps J unprojected; such a launch is for inspection only)
    bool raw_rows = false;
    if constexpr (DIAG) raw_rows = a.dbg && a.dbg_stage == 1;
    stage_rows<DIAG>(m, l, AK, S, MJ_L, !raw_rows, stamps);
    MJ_STAMP(ST_ROWS)
    if constexpr (DIAG)
      if (raw_rows) MJ_FOR(i, l.total) a.dbg[(size_t)env * l.total + i] = S[i];
    RK.dof = wv::opaque_lane(RK.dof);
    stage_pgs<DIAG, BIG>(m, l, K, RK, S, MJ_L, stamps, a.few != 0);
    MJ_STAMP(ST_PGS)
    load_euler_constants(m, a, MJ_L, EK);
    // (sensors belong to a Runge-Kutta frame's first pass, the step's own mj_forward; the later passes skip them)
    if (m.integrator == 0 || a.rk_stage == 0) {
      RecS RS;
      fetch_lane_record(a.lane_rec, REC_S, MJ_L, RS);
      stage_sensors(m, l, K, RS.SK, S, MJ_L);
    }
    MJ_STAMP(ST_SENSORS)
    if constexpr (DIAG)
      if (a.dbg && a.dbg_stage == 0) {
        MJ_FOR(i, l.total) a.dbg[(size_t)env * l.total + i] = S[i];
        wv::sync();          // (the integrator's first act is to put the inertia matrix back over its factor)
      }
    if (a.frames) {
      real* F = a.frames + (size_t)env * frame_doubles(m);
      const int* I = (const int*)(S + l.ints);
      MJ_FOR(i, 3 * m.nbody) F[i] = S[l.xpos + i];
      MJ_FOR(i, 4 * m.nbody) F[3 * m.nbody + i] = S[l.xquat + i];
      MJ_FOR(g, m.ngeom) {
        V3 gp; Quat gq;
        geom_frame(m, l, S, g, gp, gq);
        st3(F + 7 * m.nbody + 3 * g, gp);
        stq(F + 7 * m.nbody + 3 * m.ngeom + 4 * g, gq);
      }
      int ncon = I[I_NCON];
      if (L == 0) F[7 * m.nbody + 7 * m.ngeom] = ncon;
      MJ_FOR(c, m.nconmax) {
        F[7 * m.nbody + 7 * m.ngeom + 1 + 2 * c] = c < ncon ? I[l.i_cong1 + c] : -1;
        F[7 * m.nbody + 7 * m.ngeom + 2 + 2 * c] = c < ncon ? I[l.i_cong2 + c] : -1;
      }
    }
    if (a.scene) write_scene_row(m, l, S, MJ_L, a.scene + (size_t)env * scene_doubles(m));
    if (!a.forward_only) {
      RK.dof = wv::opaque_lane(RK.dof);
      if (m.integrator == 0) { mkeep_put(m, l, S, MJ_L, Mk); stage_euler(m, l, K, RK, EK, S, MJ_L); }
      else stage_rk4(m, l, S, MJ_L, a.rk_stage, a.rk + (size_t)env * (m.nq + 3 * m.nv));
    }
    if (ops_staged) {
      int* TI = (int*)(S + l.bias);
      if (L < n_act_row) S[l.bias + L] = act_reg;
      if (L < n_store_row) S[l.bias + t_store + L] = store_reg;
      if (L < 4 * a.n_op) S[l.bias + t_pf + L] = pf_reg;
      if (L < 8 * a.n_op) TI[2 * t_pi + L] = pi_reg;
      if (L < a.n_agent) { TI[2 * t_ag + L] = len_reg; TI[2 * t_ag + a.n_agent + L] = body_reg; }
      wv::sync();
    }
    MJ_STAMP(ST_EULER)
  }
  // (a launch cut at `store` ends here: cut launches write no state back, so the launches cut at successive stages
  // of tools/stage_mix.py all run on the same states)
  MJ_CUT(ST_STORE)
  // state out
  // The end of the step reads its arguments and the copy's id afresh (wv::fresh: the same values, their origin hidden
  // from the optimiser).  Otherwise every row address the prologue formed -- base pointer + copy x stride, a dozen
  // 64-bit scalars -- is kept for the stores down here, a whole step later, and waits for them in spilled registers.
  const int env_done = env, L_all = L;
  {
  const StepArgs& a = *wv::fresh(&a_in);
  const int env = wv::opaque_uniform(env_done);
  const int L = wv::opaque_lane(L_all);
  if (!a.forward_only) {
    MJ_FOR(i, m.nq) a.qpos[(size_t)env * m.nq + i] = S[l.qpos + i];
    MJ_FOR(i, m.nv) a.qvel[(size_t)env * m.nv + i] = S[l.qvel + i];
    MJ_FOR(i, m.nu) a.ctrl[(size_t)env * m.nu + i] = S[l.ctrl + i];
  }
  if (a.forward_only != 2) MJ_FOR(i, m.nv) a.warm[(size_t)env * m.nv + i] = S[l.warm + i];
  if (m.integrator == 0 || a.rk_stage == 0 || reset_only) {
    MJ_FOR(i, m.nsensordata) a.sensordata[(size_t)env * m.nsensordata + i] = S[l.sens + i];
  } else if (!a.more_frames) {        // the last pass of a Runge-Kutta frame: the observation gather reads the first pass's sensors
    MJ_FOR(i, m.nsensordata) S[l.sens + i] = a.sensordata[(size_t)env * m.nsensordata + i];
    wv::sync();
  }
  // per-agent observation gather: sensordata | qpos | qvel (sensordata is the pre-integration forward pass,
  // qpos/qvel are post-integration, exactly as the reference reads them after mj_step)
  // (observation rows: all agents' [n_env][n_agent][obs_dim] float64, or -- mjrl_set_io_layout -- one agent's
  // [n_env][obs_dim], as float when obs_f32)
  const int io_agent = a.io_agent1 - 1;
  const int obs_row = io_agent < 0 ? a.n_agent * a.obs_dim : a.obs_dim, obs_g0 = io_agent < 0 ? 0 : io_agent * a.obs_dim;
  auto obs_put = [&](size_t idx, real v) {
    if (a.obs_f32) ((float*)a.obs)[idx] = (float)v; else a.obs[idx] = v;
  };
  // slot j of agent ag's row (an agent that has no row in the buffer: nothing is written)
  auto obs_put_agent = [&](int ag, int j, real v) {
    if (io_agent < 0) obs_put(((size_t)env * a.n_agent + ag) * a.obs_dim + j, v);
    else if (ag == io_agent) obs_put((size_t)env * a.obs_dim + j, v);
  };
  if (a.obs) {
    MJ_FOR(it, obs_row) {
      int code = it < 64 ? EK.gcode[0] : (it < 128 ? EK.gcode[1] : -3);
      if (code == -3) code = a.gather[obs_g0 + it];
      // slot owned by a fused dynamics op (written below) or by the camera encoder; a reset without a step runs no op:
      // the slot reads 0 like in the reset observation of the array path (camera latents are written behind the step)
      if (code == -2 && !reset_only) continue;
      real v = 0;
      if (code >= 0) {
        int kind = code >> 24, idx = code & 0xFFFFFF;
        v = kind == 0 ? S[l.sens + idx] : (kind == 1 ? S[l.qpos + idx] : S[l.qvel + idx]);
      }
      obs_put((size_t)env * obs_row + it, v);
    }
  }
  MJ_STAMP_ONLY(ST_STORE)
  if (a.stats && a.skip_frames && L < 4) {
    const int* I = (const int*)(S + l.ints);
    a.stats[4 * (size_t)env + L] = reset_only ? 0 : I[L == 0 ? I_NCON : (L == 1 ? I_NEFC : (L == 2 ? I_NITER : I_WARN))];
  }
  if (a.overflow && !a.forward_only && a.skip_frames && !reset_only && L == 0) {
    const int warn = ((const int*)(S + l.ints))[I_WARN];
    if (warn & 1) wv::atomic_add(a.overflow + 0, 1ull);
    if (warn & 2) wv::atomic_add(a.overflow + 1, 1ull);
  }
  // The copy's work bucket for the next launch: one count, one bit.  The last thing a wave issues -- the atomics
  // return nothing, and nothing is left that could have to wait behind them.
#define MJ_FILE_WORK                                                                                         \
  if (a.lpt_count_out && L == 0) {                                                                           \
    const int* Iw = (const int*)(S + l.ints);                                                                \
    unsigned work = reset_only ? 0u : (unsigned)(Iw[I_NEFC] * Iw[I_NITER] * Iw[I_COST]);                     \
    int wb = work ? 32 - __builtin_clz(work) : 0;           /* bit length */                                 \
    if (wb > LPT_BUCKETS - 1) wb = LPT_BUCKETS - 1;                                                          \
    wv::atomic_add_noret(a.lpt_count_out + wb, 1);                                                           \
    wv::atomic_or_noret(a.lpt_mask_out + (size_t)wb * a.lpt_words + (env >> 5), 1u << (env & 31));          \
  }
#define MJ_TIMELINE                                                                        \
  if constexpr (DIAG) if (a.timeline && L == 0) {                                          \
    unsigned long long* tl = a.timeline + 3 * (size_t)wv::env_index();                     \
    tl[0] = t_begin; tl[1] = wv::realtime(); tl[2] = (unsigned long long)env;              \
  }
  if (a.forward_only || a.more_frames) {
    MJ_STAMP(ST_TAIL)
    if constexpr (DIAG) if (stamps && L < N_STAMPS) wv::atomic_add(a.stamps + L, stamps->mine);
    MJ_FILE_WORK
    MJ_TIMELINE
    return;
  }
  if (reset_only) {
    // the step of a copy that is only being reset: reward 0, flags clear, an empty data store, step counter 0, a new
    // episode -- what reset() leaves (mujoco_rl.py:291-331); the observation above is the reset observation
    if (L < a.n_agent) {
      if (a.reward) a.reward[(size_t)env * a.n_agent + L] = 0.0;
      if (a.term) a.term[(size_t)env * a.n_agent + L] = 0;
      if (a.trunc) a.trunc[(size_t)env * a.n_agent + L] = 0;
    }
    if (a.store) MJ_FOR(k, a.n_agent * a.n_slot) a.store[(size_t)env * a.n_agent * a.n_slot + k] = __builtin_nan("");
    if (a.scene && a.reset_scene) MJ_FOR(k, scene_doubles(m)) a.scene[(size_t)env * scene_doubles(m) + k] = a.reset_scene[k];
    if (L == 0) {
      a.timestep[env] = 0;
      if (a.auto_mask) a.auto_mask[env] = 0;
      if (a.episode) {
        const int ep = a.episode[env] + 1;
        a.episode[env] = ep;
        if (a.variant)
          a.variant[env] = pick_of(mix64(a.variant_seed, (unsigned long long)(a.env_base + env), 0ull, (unsigned long long)ep, 2), a.n_variant);
      }
    }
    MJ_STAMP(ST_TAIL)
    if constexpr (DIAG) if (stamps && L < N_STAMPS) wv::atomic_add(a.stamps + L, stamps->mine);
    MJ_FILE_WORK
    MJ_TIMELINE
    return;
  }
  // truncation is evaluated before the counter moves (mujoco_rl.py:279,288)
  const bool truncated = ts >= a.max_steps;
  bool ended = truncated;                     // the copy's episode ended in this step (StepArgs::auto_mask)
  MJ_FOR(ag, a.n_agent) {
    if (a.trunc) a.trunc[(size_t)env * a.n_agent + ag] = truncated;
  }
  // Rewards start at 0, terminations at false (mujoco_rl.py:262-263); the fused ops then run like the plugin loop.
  // With the program staged in LDS (or no program) LANE ag IS AGENT ag: the reward and the flag are two registers, every
  // agent's op runs at once.  The reference's loop is agent-minor and sequential, which shows in one place only -- an
  // agent hears what a lower-numbered agent said THIS step and what a higher-numbered one said last step -- and that
  // is reproduced by reading the other agent's slot before and after the writes.  (A program too large for the staging
  // area keeps its rows in HBM and runs in lane 0, one agent after the other.)
  if (a.n_op == 0 || ops_staged) {
    const int ag = L;
    const bool on = L < a.n_agent;
    real rew = 0;
    bool term = false;
    if (a.n_op > 0) {
      const real* act = a.actions ? S + l.bias : nullptr;
      real* store = S + l.bias + t_store;
      const int* TI = (const int*)(S + l.bias);
      const real* prog_f = S + l.bias + t_pf;
      const int32_t* prog_i = (const int32_t*)(TI + 2 * t_pi);
      const int32_t* obs_len = (const int32_t*)(TI + 2 * t_ag);
      const int32_t* agent_body = (const int32_t*)(TI + 2 * t_ag + a.n_agent);
      auto ref_pos = [&](int ref) {
        const int kind = ref >> 16, id = ref & 0xFFFF;
        V3 t;
        if (kind == 0) t = ld3(S + l.xpos + 3 * id) + rot(ldq(S + l.xquat + 4 * id), ld3(m.body_ipos + 3 * id));
        else { Quat tq; geom_frame(m, l, S, id, t, tq); }
        return t;
      };
      const unsigned long long genv = (unsigned long long)(a.env_base + env);
      const int me = on ? ag : 0;                       // (lanes past the agents compute on agent 0 and write nothing)
      for (int op = 0; op < a.n_op; op++) {
        const int32_t* pi = prog_i + 8 * op;
        const real* pf = prog_f + 4 * op;
        const int kind = wv::first_int(pi[0]);
        if (kind == OP_LANGUAGE) {
          const int other = me == 0 ? 1 : 0;
          const real before = other < a.n_agent ? store[other * a.n_slot + pi[2]] : 0.0;
          wv::sync();
          if (on) store[me * a.n_slot + pi[2]] = act ? (real)(long long)act[me * a.act_dim + pi[1]] : 0.0;
          wv::sync();
          const real after = other < a.n_agent ? store[other * a.n_slot + pi[2]] : 0.0;
          real heard = other < me ? after : before;
          if (heard != heard) heard = 0.0;
          if (on && a.obs) obs_put_agent(me, obs_len[me] + pi[3], heard);
        } else if (kind == OP_TARGET) {
          if (on) {
            const int body = agent_body[me];
            const int adr = a.tag_adr[pi[1]], num = a.tag_num[pi[1]];
            const unsigned long long seed = (unsigned long long)pf[2];
            real* cur_slot = store + me * a.n_slot + pi[2];
            real* inv_slot = pi[3] >= 0 ? store + me * a.n_slot + pi[3] : nullptr;
            if (*cur_slot != *cur_slot) {            // first call of the episode: choose a target, empty the inventory
              *cur_slot = (real)pick_of(mix64(seed, genv, (unsigned long long)me, ep_key | (unsigned long long)ts, 0), num);
              if (inv_slot) *inv_slot = 0.0;
            }
            int cur = (int)*cur_slot;
            V3 p = ld3(S + l.xpos + 3 * body) + rot(ldq(S + l.xquat + 4 * body), ld3(m.body_ipos + 3 * body));
            V3 t = ref_pos(a.tag_ref[adr + cur]);
            V3 d3 = p - t;
            real dist = sqrt(dot(d3, d3));
            if (dist < pf[0]) {
              if (inv_slot) *inv_slot = 1.0 - *inv_slot;
              rew += pf[1];
              cur = pick_of(mix64(seed, genv, (unsigned long long)me, ep_key | (unsigned long long)ts, 1), num);
              *cur_slot = (real)cur;
              t = ref_pos(a.tag_ref[adr + cur]);
              if (pi[5] >= 0) { V3 e3 = p - t; store[me * a.n_slot + pi[5]] = sqrt(dot(e3, e3)); }
            }
            if (a.obs) {
              const int o = obs_len[me] + pi[4];
              obs_put_agent(me, o, t.x); obs_put_agent(me, o + 1, t.y); obs_put_agent(me, o + 2, t.z);
              if (inv_slot) obs_put_agent(me, o + 3, *inv_slot);
            }
          }
        } else if (on) {
          const int body = agent_body[me];
          V3 p = ld3(S + l.xpos + 3 * body) + rot(ldq(S + l.xquat + 4 * body), ld3(m.body_ipos + 3 * body));
          V3 t = v3(0, 0, 0);
          bool have = true;
          if (pi[1] == 2) {
            const real held = store[me * a.n_slot + pi[5]];
            have = held == held;                     // (no current target yet: the op does nothing)
            if (have) t = ref_pos(a.tag_ref[a.tag_adr[pi[2]] + (int)held]);
          } else {
            t = ref_pos((pi[1] << 16) | pi[2]);
          }
          if (have) {
            V3 d3 = p - t;
            real dist = sqrt(dot(d3, d3));
            if (kind == OP_DIST_REWARD) {
              if (pi[4] == 0) {
                rew += pf[0] * (-dist);
              } else {
                real prev = store[me * a.n_slot + pi[3]];
                if (prev == prev) rew += pf[0] * (prev - dist);
              }
              if (pi[3] >= 0) store[me * a.n_slot + pi[3]] = dist;
            } else if (kind == OP_DIST_DONE) {
              term = term || dist < pf[0];
            }
          }
        }
      }
    }
    if (on) {
      if (a.reward) a.reward[(size_t)env * a.n_agent + ag] = rew;
      if (a.term) a.term[(size_t)env * a.n_agent + ag] = term;
    }
    if (a.auto_mask) ended = ended || wv::ballot(on && term) != 0ull;
  } else if (L == 0) {
    // rewards start at 0, terminations at false (mujoco_rl.py:262-263); the fused ops then run like the plugin loop
    real rew[MAX_AGENT];
    bool term[MAX_AGENT];
    for (int ag = 0; ag < a.n_agent; ag++) { rew[ag] = 0; term[ag] = false; }
    // the action row and the data-store row of this copy: staged in LDS (see the load stage) or read from HBM
    const real* act = ops_staged ? (a.actions ? S + l.bias : nullptr)
                                 : (a.actions ? a.actions + (size_t)env * a.n_agent * a.act_dim : nullptr);
    real* store = ops_staged ? S + l.bias + t_store : (a.store ? a.store + (size_t)env * a.n_agent * a.n_slot : nullptr);
    const int* TI = (const int*)(S + l.bias);
    const real* prog_f = ops_staged ? S + l.bias + t_pf : a.prog_f;
    const int32_t* prog_i = ops_staged ? (const int32_t*)(TI + 2 * t_pi) : a.prog_i;
    const int32_t* obs_len = ops_staged ? (const int32_t*)(TI + 2 * t_ag) : a.agent_obs_len;
    const int32_t* agent_body = ops_staged ? (const int32_t*)(TI + 2 * t_ag + a.n_agent) : a.agent_body;
    if (resetting && !ops_staged && store)      // (the staged copy of the row was cleared when it was fetched)
      for (int k = 0; k < a.n_agent * a.n_slot; k++) store[k] = __builtin_nan("");
    // position a target reference stands for: body -> xipos, geom -> xpos (frames of this step's forward pass)
    auto ref_pos = [&](int ref) {
      const int kind = ref >> 16, id = ref & 0xFFFF;
      V3 t;
      if (kind == 0) t = ld3(S + l.xpos + 3 * id) + rot(ldq(S + l.xquat + 4 * id), ld3(m.body_ipos + 3 * id));
      else { Quat tq; geom_frame(m, l, S, id, t, tq); }
      return t;
    };
    const unsigned long long genv = (unsigned long long)(a.env_base + env);
    for (int op = 0; op < a.n_op; op++) {
      const int32_t* pi = prog_i + 8 * op;
      const real* pf = prog_f + 4 * op;
      for (int ag = 0; ag < a.n_agent; ag++) {
        if (pi[0] == OP_LANGUAGE) {
          real utter = act ? (real)(long long)act[ag * a.act_dim + pi[1]] : 0.0;
          store[ag * a.n_slot + pi[2]] = utter;
          int other = ag == 0 ? 1 : 0;
          real heard = other < a.n_agent ? store[other * a.n_slot + pi[2]] : 0.0;
          if (heard != heard) heard = 0.0;
          if (a.obs) obs_put_agent(ag, obs_len[ag] + pi[3], heard);
        } else if (pi[0] == OP_TARGET) {
          const int body = agent_body[ag];
          const int adr = a.tag_adr[pi[1]], num = a.tag_num[pi[1]];
          const unsigned long long seed = (unsigned long long)pf[2];
          real* cur_slot = store + ag * a.n_slot + pi[2];
          real* inv_slot = pi[3] >= 0 ? store + ag * a.n_slot + pi[3] : nullptr;
          if (*cur_slot != *cur_slot) {            // first call of the episode: choose a target, empty the inventory
            *cur_slot = (real)pick_of(mix64(seed, genv, (unsigned long long)ag, ep_key | (unsigned long long)ts, 0), num);
            if (inv_slot) *inv_slot = 0.0;
          }
          int cur = (int)*cur_slot;
          V3 p = ld3(S + l.xpos + 3 * body) + rot(ldq(S + l.xquat + 4 * body), ld3(m.body_ipos + 3 * body));
          V3 t = ref_pos(a.tag_ref[adr + cur]);
          V3 d3 = p - t;
          real dist = sqrt(dot(d3, d3));
          if (dist < pf[0]) {
            if (inv_slot) *inv_slot = 1.0 - *inv_slot;
            rew[ag] += pf[1];
            cur = pick_of(mix64(seed, genv, (unsigned long long)ag, ep_key | (unsigned long long)ts, 1), num);
            *cur_slot = (real)cur;
            t = ref_pos(a.tag_ref[adr + cur]);
            if (pi[5] >= 0) { V3 e3 = p - t; store[ag * a.n_slot + pi[5]] = sqrt(dot(e3, e3)); }
          }
          if (a.obs) {
            const int o = obs_len[ag] + pi[4];
            obs_put_agent(ag, o, t.x); obs_put_agent(ag, o + 1, t.y); obs_put_agent(ag, o + 2, t.z);
            if (inv_slot) obs_put_agent(ag, o + 3, *inv_slot);
          }
        } else {
          int body = agent_body[ag];
          V3 p = ld3(S + l.xpos + 3 * body) + rot(ldq(S + l.xquat + 4 * body), ld3(m.body_ipos + 3 * body));
          V3 t;
          if (pi[1] == 2) {
            const real held = store[ag * a.n_slot + pi[5]];
            if (held != held) continue;            // no current target yet
            t = ref_pos(a.tag_ref[a.tag_adr[pi[2]] + (int)held]);
          } else {
            t = ref_pos((pi[1] << 16) | pi[2]);
          }
          V3 d3 = p - t;
          real dist = sqrt(dot(d3, d3));
          if (pi[0] == OP_DIST_REWARD) {
            if (pi[4] == 0) {
              rew[ag] += pf[0] * (-dist);
            } else {
              real prev = store[ag * a.n_slot + pi[3]];
              if (prev == prev) rew[ag] += pf[0] * (prev - dist);
            }
            if (pi[3] >= 0) store[ag * a.n_slot + pi[3]] = dist;
          } else if (pi[0] == OP_DIST_DONE) {
            term[ag] = term[ag] || dist < pf[0];
          }
        }
      }
    }
    for (int ag = 0; ag < a.n_agent; ag++) {
      if (a.reward) a.reward[(size_t)env * a.n_agent + ag] = rew[ag];
      if (a.term) a.term[(size_t)env * a.n_agent + ag] = term[ag];
      ended = ended || term[ag];
    }
    if (a.auto_mask) a.auto_mask[env] = ended;        // (lane 0 alone ran the program: it writes the flag itself)
  }
  wv::sync();
  if (a.auto_mask && L == 0 && (a.n_op == 0 || ops_staged)) a.auto_mask[env] = ended;
  if (ops_staged && a.store && L < n_store_row) a.store[(size_t)env * n_store_row + L] = S[l.bias + t_store + L];
  if (L == 0) a.timestep[env] = ts + 1;
  if (resetting && a.episode && L == 0) {          // a new episode (and a new level variant), as mjrl_reset counts / chooses it
    const int ep = a.episode[env] + 1;
    a.episode[env] = ep;
    if (a.variant)
      a.variant[env] = pick_of(mix64(a.variant_seed, (unsigned long long)(a.env_base + env), 0ull, (unsigned long long)ep, 2), a.n_variant);
  }
  MJ_STAMP(ST_TAIL)
  if constexpr (DIAG) if (stamps && L < N_STAMPS) wv::atomic_add(a.stamps + L, stamps->mine);
  MJ_FILE_WORK
  MJ_TIMELINE
  }
#undef MJ_TIMELINE
#undef MJ_FILE_WORK
#undef MJ_STAMP
#undef MJ_STAMP_ONLY
#undef MJ_CUT
#undef MJ_FOR
#undef MJ_L
}

#if defined(__HIPCC__)
// The step arguments where the dispatch packet left them.  A kernel that names its by-value StepArgs parameter gets every
// field it uses loaded into scalar registers in its entry block (that is how kernel arguments are lowered) -- some forty
// pointers and twenty sizes, the whole scalar register file, live from the first instruction to wherever each is used
// and spilled to vector-register lanes in between (v_writelane / v_readlane: vector-issue slots in a kernel bound by
// vector issue).  Read through the kernarg segment pointer instead, a field is a scalar load from constant memory where
// it is used.  `offset`: byte offset of the StepArgs parameter in the kernel's argument list.
__device__ __forceinline__ const StepArgs* kernarg_step_args(int offset) {
  const char __attribute__((address_space(4)))* base =
      (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
  return (const StepArgs*)(const StepArgs __attribute__((address_space(4)))*)(base + offset);
}
#endif

// Where the StepArgs parameter of the step kernels -- all of the form kernel(pointer, StepArgs) -- sits in the kernarg
// segment: behind the pointer, rounded up to the struct's alignment.  The kernels read their arguments THERE and never
// name the by-value parameter, so nothing but these lines ties the offset to the signature.
static_assert(sizeof(void*) == 8, "the step kernels' first parameter is one 8-byte pointer");
static_assert(alignof(StepArgs) == 8, "a member with a wider alignment moves StepArgs in the kernarg segment");
constexpr int STEP_ARGS_KERNARG_OFFSET = (int)((sizeof(void*) + alignof(StepArgs) - 1) / alignof(StepArgs) * alignof(StepArgs));

// What a code object built apart from the library (a model-specialised step kernel, csrc/_spec/*.hsaco) must agree with
// the library on before it may be launched: the layout of StepArgs -- its size and the offsets of fields spread over it,
// folded into one word -- and the kernel sources both were built from (MJRL_SOURCE_DIGEST: the first 60 bits of the
// SHA-1 of the header set, handed to both builds by csrc/Makefile and kernel_cache.py).  mjrl_load_kernel compares the
// object's `mjrl_spec_abi` with the library's; a stale or mis-laid-out object is refused, not timed.
#ifndef MJRL_SOURCE_DIGEST
#define MJRL_SOURCE_DIGEST 0ull
#endif
constexpr unsigned long long step_args_layout() {
  unsigned long long h = 1469598103934665603ull;
#define MJ_LAYOUT_FIELD(f) h = (h ^ (unsigned long long)__builtin_offsetof(StepArgs, f)) * 1099511628211ull;
  MJ_LAYOUT_FIELD(qpos) MJ_LAYOUT_FIELD(timestep) MJ_LAYOUT_FIELD(actions) MJ_LAYOUT_FIELD(scatter_mode)
  MJ_LAYOUT_FIELD(gather) MJ_LAYOUT_FIELD(obs) MJ_LAYOUT_FIELD(trunc) MJ_LAYOUT_FIELD(n_env) MJ_LAYOUT_FIELD(more_frames)
  MJ_LAYOUT_FIELD(dbg) MJ_LAYOUT_FIELD(forward_only) MJ_LAYOUT_FIELD(reset_mask) MJ_LAYOUT_FIELD(auto_mask)
  MJ_LAYOUT_FIELD(reset_sens) MJ_LAYOUT_FIELD(rk) MJ_LAYOUT_FIELD(stats) MJ_LAYOUT_FIELD(stamps) MJ_LAYOUT_FIELD(prog_i)
  MJ_LAYOUT_FIELD(n_slot) MJ_LAYOUT_FIELD(store) MJ_LAYOUT_FIELD(tag_ref) MJ_LAYOUT_FIELD(env_base) MJ_LAYOUT_FIELD(variant)
  MJ_LAYOUT_FIELD(variant_seed) MJ_LAYOUT_FIELD(frames) MJ_LAYOUT_FIELD(scene) MJ_LAYOUT_FIELD(few) MJ_LAYOUT_FIELD(io_agent1) MJ_LAYOUT_FIELD(obs_f32) MJ_LAYOUT_FIELD(lane_rec)
  MJ_LAYOUT_FIELD(lpt_count_in) MJ_LAYOUT_FIELD(lpt_mask_clear) MJ_LAYOUT_FIELD(lpt_words) MJ_LAYOUT_FIELD(overflow)
  MJ_LAYOUT_FIELD(timeline) MJ_LAYOUT_FIELD(stop_after)
#undef MJ_LAYOUT_FIELD
  return (h ^ (unsigned long long)sizeof(StepArgs)) * 1099511628211ull;
}
enum { SPEC_ABI_WORDS = 3 };
#define MJRL_SPEC_ABI_INIT { mj::step_args_layout(), (unsigned long long)(MJRL_SOURCE_DIGEST), (unsigned long long)mj::STEP_ARGS_KERNARG_OFFSET }

#if defined(__HIPCC__)
// The check the production kernels do not carry: are the arguments where kernarg_step_args looks for them?  Run once per
// kernel (mjrl_create for the library's, mjrl_load_kernel for a code object's) by a kernel of the same signature plus a
// result word; compares fields spread over the struct with the by-value copy the compiler hands out.
__device__ __forceinline__ void kernarg_selfcheck(const StepArgs& by_value, int* ok) {
  const StepArgs* k = kernarg_step_args(STEP_ARGS_KERNARG_OFFSET);
  const bool same = k->qpos == by_value.qpos && k->timestep == by_value.timestep && k->obs == by_value.obs &&
                    k->n_env == by_value.n_env && k->max_steps == by_value.max_steps && k->stats == by_value.stats &&
                    k->few == by_value.few && k->lpt_words == by_value.lpt_words && k->overflow == by_value.overflow &&
                    k->stop_after == by_value.stop_after;
  if (threadIdx.x == 0) *ok = same ? 1 : 0;
}
#endif

// the diagnostic build under its historical name (the CPU emulation under tests/emu steps copies through this one)
__device__ inline void env_step(const DevModel& m, const StepArgs& a, real* S) { env_step_t<true>(m, a, S); }

}  // namespace mj

#endif
