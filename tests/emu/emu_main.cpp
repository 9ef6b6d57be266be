// TEST INFRASTRUCTURE: runs the device step code (csrc/mjrl_step.h, unmodified) on the CPU, one env copy,
// 64 lanes as fibers (see emu_wave.h).  Built into tests/emu/_build/libmjrl_emu.so by tests/emu/Makefile and
// used by the CPU test-suite to check the kernel logic against the oracle before any GPU time is spent.
// It is not a product path: the Python package only ever loads libmjrl_hip.so.
#include "emu_wave.h"

#ifdef MJRL_EMU_SANITIZED
#include <sanitizer/asan_interface.h>
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../mujoco-rl-environment-wrapper_amd/csrc/mjrl_step.h"

// Lane switch: callee-saved registers and the stack pointer, nothing else -- no system call (swapcontext saves and
// restores the signal mask with one on every switch, which was most of the emulation's run time).
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.hidden emu_switch
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch, .-emu_switch
)");

namespace emu {
int cur_lane = 0, cur_env = 0;
long sync_count[64];
double xd[64];
long long xi[64];
static void* main_sp;
static void* lane_sp[64];
static bool done[64];
static std::vector<char> stacks;

static int next_live(int from) {
  for (int k = 1; k <= 64; k++) {
    int i = (from + k) % 64;
    if (!done[i]) return i;
  }
  return -1;
}

void yield_lane() {
  int me = cur_lane, nxt = next_live(me);
  if (nxt < 0 || nxt == me) return;
  cur_lane = nxt;
  emu_switch(&lane_sp[me], lane_sp[nxt]);
}

struct Job { const DevModel* m; const mj::StepArgs* a; double* lds; int* lookup_out; int lookup_wg; };
static Job job;

static void lane_entry() {
  if (job.lookup_out) {          // the dispatch lookup alone (emu_lpt_lookup)
    const int L = wv::lane();
    int bucket = 0;
    const int my_count = job.a->lpt_count_in[L & (mj::LPT_BUCKETS - 1)];
    const int copy = mj::lpt_copy_of(*job.a, L, job.lookup_wg, my_count, bucket);
    if (L == 0) *job.lookup_out = copy;
  } else {
    mj::env_step(*job.m, *job.a, job.lds);
  }
  int me = cur_lane;
  done[me] = true;
  int nxt = next_live(me);
  void* dead;
  if (nxt < 0) emu_switch(&dead, main_sp);        // the last lane hands control back to run_wave; never resumed
  cur_lane = nxt;
  emu_switch(&dead, lane_sp[nxt]);
  __builtin_trap();
}

static int run_wave(const DevModel& m, const mj::StepArgs& a, double* lds, int* lookup_out = nullptr, int lookup_wg = 0) {
  const size_t stack_bytes = 1 << 20;
  stacks.resize(64 * stack_bytes + 64);
#ifdef MJRL_EMU_SANITIZED
  // the fibers of the previous wave never returned: their frames' redzones are still poisoned
  __asan_unpoison_memory_region(stacks.data(), stacks.size());
#endif
  job = {&m, &a, lds, lookup_out, lookup_wg};
  for (int i = 0; i < 64; i++) {
    done[i] = false;
    sync_count[i] = 0;
    // a fresh lane: six zeroed callee-saved registers, then the entry point where emu_switch's `ret` finds it, on a
    // stack that is 16-byte aligned at the entry's (virtual) call site
    uintptr_t top = ((uintptr_t)(stacks.data() + (size_t)(i + 1) * stack_bytes)) & ~(uintptr_t)15;
    void** sp = (void**)(top - 16);
    sp[1] = nullptr;                         // return address of lane_entry (it never returns)
    sp[0] = (void*)lane_entry;
    for (int k = 1; k <= 6; k++) sp[-k] = nullptr;
    lane_sp[i] = (void*)(sp - 6);
  }
  cur_lane = 0;
  emu_switch(&main_sp, lane_sp[0]);
  for (int i = 1; i < 64; i++)
    if (sync_count[i] != sync_count[0]) {
      fprintf(stderr, "emu: lane %d passed %ld barriers, lane 0 passed %ld (divergent barrier)\n", i, sync_count[i], sync_count[0]);
      return 1;
    }
  return 0;
}
}  // namespace emu

extern "C" {

int emu_lds_total(const void* blob, size_t nbytes) {
  DevModel m;
  if (mjrl_model_from_blob(&m, blob, nbytes, blob)) return -1;
  mj::Lay l;
  mj::make_layout(m, l);
  return l.total;
}

int emu_lds_offset(const void* blob, size_t nbytes, const char* region) {
  DevModel m;
  if (mjrl_model_from_blob(&m, blob, nbytes, blob)) return -1;
  mj::Lay l;
  mj::make_layout(m, l);
#define R(name) if (!strcmp(region, #name)) return l.name;
  R(qpos) R(qvel) R(ctrl) R(warm) R(xpos) R(xquat) R(xanchor) R(xaxis) R(com) R(cinert) R(crb) R(cdof) R(cdofdot)
  R(cvel) R(cacc) R(LD) R(Dinv) R(gpos) R(gquat) R(bias) R(smooth) R(qaccs) R(x) R(qfc) R(qacc) R(con) R(J) R(row)
  R(sens) R(ints) R(total) R(ldj) R(i_item) R(i_cong1) R(i_cong2) R(i_conadr) R(i_rowid) R(i_rowinfo)
#undef R
  return -1;
}

// in-launch reset of the next emu_step call's first frame (mjrl_set_step_reset_mask): flag != 0 and the reset image's
// warm start; cleared by the call
// (flag 2: a reset without a step; reset_sens: the reset image's sensor readings, needed by that kind)
static unsigned char g_reset_flag = 0;
static const double *g_reset_warm = nullptr, *g_reset_sens = nullptr;
void emu_set_step_reset(int flag, const double* reset_warm) { g_reset_flag = (unsigned char)(flag < 0 ? 0 : flag); g_reset_warm = reset_warm; }
void emu_set_reset_sens(const double* reset_sens) { g_reset_sens = reset_sens; }
// autoreset kept by the step itself (mjrl_set_autoreset): the copy's flag byte (read and written by every step) and the mode
static unsigned char* g_auto_mask = nullptr;
static int g_auto_mode = 0;
static int* g_episode = nullptr;
void emu_set_autoreset(unsigned char* mask_byte, int mode, int* episode) { g_auto_mask = mask_byte; g_auto_mode = mode; g_episode = episode; }

// object-tag tables and the global id of the copy for the following emu_step calls (mjrl_set_tag_tables / mjrl_set_env_base)
static const int32_t *g_tag_adr = nullptr, *g_tag_num = nullptr, *g_tag_ref = nullptr;
static int g_env_base = 0;
void emu_set_tags(const int32_t* adr, const int32_t* num, const int32_t* ref, int env_base) {
  g_tag_adr = adr; g_tag_num = num; g_tag_ref = ref; g_env_base = env_base;
}

// StepArgs::few of the following emu_step calls: the solver forms of the build for batches of at most one wave per SIMD
static int g_few = 0;
void emu_set_few(int few) { g_few = few; }

// I/O layout of the following emu_step calls (mjrl_set_io_layout): agent + 1 (0: every agent's rows), float32 observations
static int g_io_agent1 = 0, g_obs_f32 = 0;
void emu_set_io_layout(int agent1, int obs_f32) { g_io_agent1 = agent1; g_obs_f32 = obs_f32; }

// longest-first dispatch tables of the following emu_step calls (launch_step's lpt_count_in / lpt_list_in); null: identity
static const int* g_lpt_count = nullptr;
static const unsigned* g_lpt_mask = nullptr;
static int g_lpt_words = 0;
void emu_set_lpt(const int* count_in, const unsigned* mask_in, int words) { g_lpt_count = count_in; g_lpt_mask = mask_in; g_lpt_words = words; }

// the copy workgroup ids[k] of a launch would step, given dispatch tables (mj::lpt_copy_of run by a whole emulated wave)
int emu_lpt_lookup(const int* count_in, const unsigned* mask_in, int words, int n_env, const int* ids, int n, int* out) {
  DevModel m{};
  mj::StepArgs a{};
  a.lpt_count_in = count_in; a.lpt_mask_in = mask_in; a.lpt_words = words; a.n_env = n_env;
  for (int k = 0; k < n; k++)
    if (emu::run_wave(m, a, nullptr, out + k, ids[k])) return 2;
  return 0;
}

// one env copy, `nsteps` step() calls with the same actions; state arrays are updated in place
int emu_step(const void* blob, size_t nbytes, double* qpos, double* qvel, double* ctrl, double* warm, double* sens,
             int* timestep, const double* actions, const int32_t* scatter, int n_agent, int act_dim, int scatter_mode,
             const int32_t* gather, int obs_dim, double* obs, int skip_frames, int nsteps, int max_steps, double* dbg,
             int dbg_stage, int forward_only, const int32_t* prog_i, const double* prog_f, int n_op, int n_slot,
             const int32_t* agent_body, const int32_t* agent_obs_len, double* store, double* reward,
             unsigned char* term, unsigned char* trunc) {
  DevModel m;
  if (mjrl_model_from_blob(&m, blob, nbytes, blob)) return -1;
  mj::Lay l;
  mj::make_layout(m, l);
  std::vector<double> lds(l.total, 0.0);
  std::vector<double> rk(m.nq + 3 * m.nv + 1, 0.0);           // the per-copy Runge-Kutta scratch of the real launch
  mj::StepArgs a{};
  a.rk = rk.data();
  a.qpos = qpos; a.qvel = qvel; a.ctrl = ctrl; a.warm = warm; a.sensordata = sens; a.timestep = timestep;
  a.actions = actions; a.scatter = scatter; a.n_agent = n_agent; a.act_dim = act_dim; a.scatter_mode = scatter_mode;
  a.gather = gather; a.obs_dim = obs_dim; a.obs = obs;
  a.reward = reward; a.term = term; a.trunc = trunc;
  a.prog_i = prog_i; a.prog_f = prog_f; a.n_op = forward_only ? 0 : n_op; a.n_slot = n_slot;
  a.agent_body = agent_body; a.agent_obs_len = agent_obs_len; a.store = store;
  a.lpt_count_in = g_lpt_count; a.lpt_mask_in = g_lpt_mask; a.lpt_words = g_lpt_words;
  a.tag_adr = g_tag_adr; a.tag_num = g_tag_num; a.tag_ref = g_tag_ref; a.env_base = g_env_base;
  a.max_steps = max_steps; a.n_env = 1; a.few = g_few;
  a.io_agent1 = g_io_agent1; a.obs_f32 = g_obs_f32;
  std::vector<int32_t> lane_rec((size_t)mj::LANE_REC_INTS, 0);        // (mjrl_create builds this table once per handle)
  mj::build_lane_records(m, l, lane_rec.data());
  a.lane_rec = lane_rec.data();
  a.dbg = dbg; a.dbg_stage = dbg_stage; a.forward_only = forward_only;
  emu::cur_env = 0;
  // one frame per wave run, like the launches of launch_step in mjrl_capi.hip
  for (int s = 0; s < nsteps; s++) {
    const int passes = (m.integrator == 1 && !forward_only) ? 4 : 1;
    int launches = skip_frames > 0 ? skip_frames * passes : 1;
    for (int f = 0; f < launches; f++) {
      mj::StepArgs b = a;
      b.rk_stage = f % passes;
      b.skip_frames = skip_frames > 0 ? 1 : 0;
      b.more_frames = f < launches - 1;
      if (f > 0) b.scatter = nullptr;
      b.first_frame = f == 0;
      if (b.more_frames) { b.dbg = nullptr; b.frames = nullptr; }
      b.reset_warm = g_reset_warm; b.reset_sens = g_reset_sens;
      if (s == 0 && g_reset_flag) b.reset_mask = &g_reset_flag;
      if (!forward_only) { b.auto_mask = g_auto_mask; b.auto_mode = g_auto_mode; b.episode = g_episode; }
      if (emu::run_wave(m, b, lds.data())) return 2;
    }
  }
  g_reset_flag = 0;
  return 0;
}
}
