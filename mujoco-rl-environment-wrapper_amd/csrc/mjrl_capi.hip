// libmjrl_hip.so: kernels + the C-ABI declared in include/mjrl.h.
// One workgroup = one 64-lane wavefront = one env copy; the env's working set lives in dynamic LDS.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mjrl.h"
#include "mjrl_encoder.h"
#include "mjrl_rayf.h"
#include "mjrl_step.h"

namespace {

// The model descriptor (sizes + ~100 section pointers) is read through a pointer: passed by value it would sit in
// SGPRs for the whole kernel and spill to VGPR lanes (2600 v_readlane/v_writelane in the ISA); behind a const
// __restrict__ pointer every field is a scalar load at its point of use.
// Two kinds (stage_pgs, BIG): mjrl_step_kernel without the solver forms that need more than 256 registers and held to 256
// itself (with every size a run-time value the body wants 336: capped, it spills a hundred registers to scratch and
// still runs the 2-agent level at 18.7 M env-steps/s against 13.4 at one wave per SIMD); mjrl_step_kernel_big with those
// forms and no cap, for models whose LDS image allows one wave per SIMD anyway (mj::pgs_roomy).
#define MJRL_TWO_WAVES __attribute__((amdgpu_waves_per_eu(2, 2)))
__global__ __launch_bounds__(64) MJRL_TWO_WAVES void mjrl_step_kernel(const DevModel* __restrict__ mp, mj::StepArgs a) {
  extern __shared__ double lds[];
  mj::env_step_t<false, false>(*mp, *mj::kernarg_step_args(mj::STEP_ARGS_KERNARG_OFFSET), lds);      // (the arguments where the packet left them: mjrl_step.h)
}
__global__ __launch_bounds__(64) void mjrl_step_kernel_big(const DevModel* __restrict__ mp, mj::StepArgs a) {
  extern __shared__ double lds[];
  mj::env_step_t<false, true>(*mp, *mj::kernarg_step_args(mj::STEP_ARGS_KERNARG_OFFSET), lds);
}
// The same step with the diagnostics compiled in (stage clock, wave timeline, LDS dump, stage cuts): the launches of
// mjrl_step_debug / _profile / _timeline / _truncated and the LDS read-back of mjrl_query.  Same arithmetic, same bits.
__global__ __launch_bounds__(64) MJRL_TWO_WAVES void mjrl_step_kernel_diag(const DevModel* __restrict__ mp, mj::StepArgs a) {
  extern __shared__ double lds[];
  // (the diagnostic build checks what the production build assumes: the arguments sit 8 bytes into the kernarg segment)
  const mj::StepArgs* k = mj::kernarg_step_args(mj::STEP_ARGS_KERNARG_OFFSET);
  if (k->qpos != a.qpos || k->n_env != a.n_env || k->lpt_words != a.lpt_words) __builtin_trap();
  mj::env_step_t<true, false>(*mp, *k, lds);
}
__global__ __launch_bounds__(64) void mjrl_step_kernel_big_diag(const DevModel* __restrict__ mp, mj::StepArgs a) {
  extern __shared__ double lds[];
  const mj::StepArgs* k = mj::kernarg_step_args(mj::STEP_ARGS_KERNARG_OFFSET);
  if (k->qpos != a.qpos || k->n_env != a.n_env || k->lpt_words != a.lpt_words) __builtin_trap();
  mj::env_step_t<true, true>(*mp, *k, lds);
}

// Are the arguments where the production kernels read them (mj::kernarg_selfcheck)?  Same signature as the step kernels
// plus a result word; run once per process by mjrl_create.
__global__ __launch_bounds__(64) void mjrl_selfcheck_kernel(const DevModel* __restrict__ mp, mj::StepArgs a, int* ok) {
  (void)mp;
  mj::kernarg_selfcheck(a, ok);
}
const unsigned long long LIB_ABI[mj::SPEC_ABI_WORDS] = MJRL_SPEC_ABI_INIT;

// Masked reset of the HBM state: mj_resetData + mj_forward (mujoco_parent.py:349-350) for the selected copies.  Every
// copy resets to the same state, so what mj_forward leaves there -- the warm start and the sensor readings -- is computed
// once per model (reset image, mjrl_create) and copied; copies that are not selected are not touched at all.  With
// `obs` the reset observations (mujoco_rl.py:314: sensordata | qpos | qvel per agent; slots owned by fused dynamics: 0)
// of the selected copies are gathered from the image.
__global__ void mjrl_reset_kernel(DevModel m, double* qpos, double* qvel, double* ctrl, double* warm, double* sens,
                                  int* timestep, const unsigned char* mask, int n_env, double* store, int store_per_env,
                                  const double* warm0, const double* sens0, const int32_t* gather, int n_agent, int obs_dim,
                                  double* obs, int* variant, int* episode, int n_variant, unsigned long long variant_seed,
                                  int env_base, unsigned char* auto_mask, double* scene, const double* scene0, int io_agent,
                                  int obs_f32) {
  int env = blockIdx.x;
  if (env >= n_env || (mask && !mask[env])) return;
  // (the ray caster's scene row: the frames mj_forward leaves at the reset state)
  if (scene && scene0)
    for (int i = threadIdx.x; i < mj::scene_doubles(m); i += blockDim.x) scene[(size_t)env * mj::scene_doubles(m) + i] = scene0[i];
  if (auto_mask && threadIdx.x == 0) auto_mask[env] = 0;      // (a reset by hand settles a pending autoreset)
  // A new episode.  The episode count of a copy always moves (it is part of the key of every on-device random choice,
  // so that episodes differ from one another the way the reference's random.randint draws do); with level variants on,
  // the copy also gets a new variant (random.choice, mujoco_parent.py:352).
  if (episode && threadIdx.x == 0) {
    const int ep = episode[env] + 1;
    episode[env] = ep;
    if (variant)
      variant[env] = mj::pick_of(mj::mix64(variant_seed, (unsigned long long)(env_base + env), 0ull, (unsigned long long)ep, 2), n_variant);
  }
  // an empty data store (mujoco_rl.py:312): every slot "absent"
  for (int i = threadIdx.x; i < store_per_env; i += blockDim.x) store[(size_t)env * store_per_env + i] = __longlong_as_double(0x7FF8000000000000ll);
  for (int i = threadIdx.x; i < m.nq; i += blockDim.x) qpos[(size_t)env * m.nq + i] = m.qpos0[i];
  for (int i = threadIdx.x; i < m.nv; i += blockDim.x) { qvel[(size_t)env * m.nv + i] = 0; warm[(size_t)env * m.nv + i] = warm0 ? warm0[i] : 0.0; }
  for (int i = threadIdx.x; i < m.nu; i += blockDim.x) ctrl[(size_t)env * m.nu + i] = 0;
  if (sens0) for (int i = threadIdx.x; i < m.nsensordata; i += blockDim.x) sens[(size_t)env * m.nsensordata + i] = sens0[i];
  if (threadIdx.x == 0) timestep[env] = 0;
  if (obs && gather) {
    // (rows in the layout of the step's: every agent's as float64, or one agent's / as float -- mjrl_set_io_layout)
    const int row = io_agent < 0 ? n_agent * obs_dim : obs_dim, g0 = io_agent < 0 ? 0 : io_agent * obs_dim;
    for (int it = threadIdx.x; it < row; it += blockDim.x) {
      int code = gather[g0 + it];
      double v = 0;
      if (code >= 0) {
        int kind = code >> 24, idx = code & 0xFFFFFF;
        v = kind == 0 ? (sens0 ? sens0[idx] : 0.0) : (kind == 1 ? m.qpos0[idx] : 0.0);
      }
      if (obs_f32) ((float*)obs)[(size_t)env * row + it] = (float)v;
      else obs[(size_t)env * row + it] = v;
    }
  }
}

// Fixed body-mounted cameras as a ray caster (replaces mjv_updateScene + mjr_render + mjr_readPixels of
// mujoco_parent.py:518-538 for the agent cameras).  Camera convention of the reference's renderer: looks along -z, +x
// right, +y up, vertical field of view fovy, rows stored bottom-up (glReadPixels order), uint8 RGB.
//
// Two kernels.  mjrl_camera_frames_kernel, one wave per env copy: body and geom frames from the current qpos (the step
// kernel's kinematics), written as a scene row  [geom pos 3G | geom matrix 9G | camera pos 3C | camera matrix 9C |
// light pos 3N | light dir 3N]  to HBM.
// mjrl_render_kernel, grid (n_env, ncam * tiles): a wave renders a group of 8x8 pixel blocks of one camera of one copy
// from that row.  (Round 1 let every render wave redo the copy's kinematics behind the 20 KB step image: 16 tiles x 2
// cameras repeated it 32 times per copy and the image held a CU to 7 waves.)
using mj::scene_doubles;
// x^y for x in [0, 1], y >= 0 as exp2(y log2 x) on the transcendental unit (v_log_f32, v_exp_f32): three instructions
// where the library's powf -- also behind __powf -- is 170; a tenth of a colour level at worst after the 255 scaling
__device__ __forceinline__ float fast_pow(float x, float y) {
  return y > 0.0f ? __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)) : 1.0f;
}
#ifndef MJRL_RENDER_VARIANT
#define MJRL_RENDER_VARIANT 0        // (experiments: 1 = round 2's shading, 2 = byte stores, 3 = no type-specific tests, 4 = one candidate)
#endif
// one light of the ray kernel in LDS (floats): camera-relative position 3 | direction 3 | attenuation 3 | cos(cutoff) |
// exponent | ambient 3 | diffuse 3 | specular 3 | directional | casts a shadow
enum { LIGHT_FLOATS = 22 };
// minimum / maximum of a float over the wave, in every lane (DPP inside the rows of 16, v_readlane across them)
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_minf(float v) {
  v = fminf(v, dppf<0xB1>(v)); v = fminf(v, dppf<0x4E>(v)); v = fminf(v, dppf<0x141>(v)); v = fminf(v, dppf<0x140>(v));
  auto at = [&](int l) { return __int_as_float(wv::lane_int(__float_as_int(v), l)); };
  return fminf(fminf(at(0), at(16)), fminf(at(32), at(48)));
}
__device__ __forceinline__ float wave_maxf(float v) { return -wave_minf(-v); }

__global__ __launch_bounds__(64) void mjrl_camera_frames_kernel(DevModel m, const double* qpos, int n_env, double* scene) {
  extern __shared__ double lds[];
  using namespace mj;
  const int L = wv::lane(), env = blockIdx.x;
  real* S = lds;
  Lay l;
  make_layout(m, l);
  LaneK K;
  load_lane_constants(m, L, K);
  TabRegs TR;
  tab_issue(m, L, TR);
  KinK KK;
  load_kin_constants(m, L, K, KK);
  for (int i = L; i < m.nq; i += 64) S[l.qpos + i] = qpos[(size_t)env * m.nq + i];
  stage_constants(m, l, S, L, TR);
  wv::sync();
  stage_kinematics(m, l, K, KK, S, L);
  write_scene_row(m, l, S, L, scene + (size_t)env * scene_doubles(m));
}

// (Single precision: the kernel is bound by vector-instruction issue -- 93 M VALU instructions per 512 x 2 cameras,
// profiles/r02_pmc_render.txt -- and fp32 issues at twice the fp64 rate.  The tests bound the pixels that may differ from
// the oracle's fp64 ray caster.)
// The ray caster proper: blocks blk0, blk0 + blk_step, .. (< blk1) of camera `cam` of the copy whose scene row is `row`, drawn by the
// calling wave into `img` (rows of row_bytes bytes), its tables staged in the wave's LDS `ldsf`.
__device__ __forceinline__ void render_body(const DevModel& m, const double* row, int cam, int width, int height, int blk0,
                                            int blk1, int blk_step, float* ldsf, unsigned char* img, int row_bytes,
                                            const double* rgba_tab, const float* consts, int loose_culls) {
  using namespace mj;
  const int L = wv::lane(), tid = L, nthr = 64;
  auto barrier = [&]() { wv::sync(); };
  // LDS: the copy's geom positions, matrices and sizes
  float* GP = ldsf;
  float* GM = ldsf + 3 * m.ngeom;
  float* GS = ldsf + 12 * m.ngeom;
  float* MP = ldsf + 15 * m.ngeom;            // material properties: specular, 128 x shininess, emission
  float* GC = ldsf + 18 * m.ngeom;            // the copy's geom colours (its level variant's)
  float* LT = ldsf + 21 * m.ngeom;            // light table: entry 0 the headlight, then the level's lights
  for (int i = tid; i < 12 * m.ngeom; i += nthr) ldsf[i] = (float)row[i];
  for (int i = tid; i < 3 * m.ngeom; i += nthr) {
    GS[i] = (float)m.geom_size[i];
    MP[i] = (float)m.geom_matprop[i] * (i % 3 == 1 ? 128.0f : 1.0f);
  }
  // the camera, and every geom's position relative to it (the subtraction in double: positions are metres from the
  // arena's origin, differences are what the rays see)
  const double cpx = row[12 * m.ngeom + 3 * cam], cpy = row[12 * m.ngeom + 3 * cam + 1], cpz = row[12 * m.ngeom + 3 * cam + 2];
  // (the camera matrix lives in LDS, behind the light table: nine wave-uniform floats would otherwise sit in nine vector
  // registers for the whole kernel -- there is no scalar float unit to keep them in -- and cost a wave per SIMD)
  float* cm = LT + LIGHT_FLOATS * (m.nlight + 1);
  if (tid < 9) cm[tid] = (float)row[12 * m.ngeom + 3 * m.ncam + 9 * cam + tid];
  barrier();
  for (int g = tid; g < m.ngeom; g += nthr) {       // camera-relative positions
    GP[3 * g] = (float)(row[3 * g] - cpx); GP[3 * g + 1] = (float)(row[3 * g + 1] - cpy); GP[3 * g + 2] = (float)(row[3 * g + 2] - cpz);
  }
  // the copy's colours: its level variant's (mjrl_set_variants), else the model's
  auto rgba_of = [&](int g, int k) { return (float)(rgba_tab ? rgba_tab[4 * g + k] : (double)m.geom_rgba[4 * g + k]); };
  for (int i = tid; i < 3 * m.ngeom; i += nthr) GC[i] = rgba_of(i / 3, i % 3);
  if (tid == 0) {
    // the headlight (visual/headlight) as entry 0: directional, shining along the viewing direction (L = V = the
    // camera's +z axis); an inactive one contributes nothing
    float* T = LT;
    const float on = m.headlight[0] != 0 ? 1.0f : 0.0f;
    T[0] = T[1] = T[2] = 0.0f;
    T[3] = -cm[2]; T[4] = -cm[5]; T[5] = -cm[8];
    T[6] = 1.0f; T[7] = T[8] = 0.0f; T[9] = -2.0f; T[10] = 0.0f;
    for (int k = 0; k < 9; k++) T[11 + k] = on * (float)m.headlight[1 + k];
    T[20] = 1.0f;
    T[21] = 0.0f;                 // (the headlight casts no shadow)
  }
  for (int li = tid; li < m.nlight; li += nthr) {   // the level's lights, camera-relative
    float* T = LT + LIGHT_FLOATS * (li + 1);
    const double* lp = row + 12 * m.ngeom + 12 * m.ncam + 3 * li;
    const double* ld = row + 12 * m.ngeom + 12 * m.ncam + 3 * m.nlight + 3 * li;
    T[0] = (float)(lp[0] - cpx); T[1] = (float)(lp[1] - cpy); T[2] = (float)(lp[2] - cpz);
    for (int k = 0; k < 3; k++) {
      T[3 + k] = (float)ld[k];
      T[6 + k] = (float)m.light_attenuation[3 * li + k];
      T[11 + k] = (float)m.light_ambient[3 * li + k];
      T[14 + k] = (float)m.light_diffuse[3 * li + k];
      T[17 + k] = (float)m.light_specular[3 * li + k];
    }
    T[9] = consts[m.ncam + li];
    T[10] = (float)m.light_exponent[li];
    T[20] = m.light_directional[li] ? 1.0f : 0.0f;
    T[21] = m.light_castshadow[li] ? 1.0f : 0.0f;
  }
  barrier();
  // Boxes (the arena's walls: 20 m long, bounding spheres of 10 m -- inside every block's cone) get a tighter test than
  // the sphere's: the block's rays lie in the frustum of its four edge planes, all through the camera, with inward
  // normals X + a Z, -X - b Z, Y + c Z, -Y - d Z (X, Y, Z the camera's axes; a, b / c, d the block's left, right /
  // bottom, top edges per unit depth), and a box wholly behind one of them is seen by no ray of the block.  Per box and
  // camera axis W: the centre's coordinate rel . W and the three half-extents (u_i . W) s_i -- twelve floats, here.
  float* BX = cm + 12;
  // (a geom's type and bounding radius, rounded up, as the culls read them -- from LDS per block, not from five registers
  // per lane for the whole kernel: the pair kernel is held to 128 and spilled them)
  float* GK = BX + 12 * m.ngeom;
  for (int g = tid; g < m.ngeom; g += nthr) {
    const bool seen = (rgba_tab ? rgba_tab[4 * g + 3] : (double)m.geom_rgba[4 * g + 3]) != 0;
    GK[2 * g] = seen ? (float)m.geom_type[g] : -1.0f;
    GK[2 * g + 1] = (float)m.geom_rbound[g] * (1.0f + 1e-5f) + 1e-5f;
  }
  for (int g = tid; g < m.ngeom; g += nthr) {
    const F3 rel = ldf3(GP + 3 * g), sz = ldf3(GS + 3 * g);
    for (int w = 0; w < 3; w++) {
      const F3 W = colf(cm, w);
      BX[12 * g + 4 * w] = dotf(rel, W);
      BX[12 * g + 4 * w + 1] = dotf(colf(GM + 9 * g, 0), W) * sz.x;
      BX[12 * g + 4 * w + 2] = dotf(colf(GM + 9 * g, 1), W) * sz.y;
      BX[12 * g + 4 * w + 3] = dotf(colf(GM + 9 * g, 2), W) * sz.z;
    }
    // (a plane keeps its normal in camera coordinates there: a block whose four edge rays all point away from the
    // plane's front -- the half of the image above the horizon -- cannot see it)
    if (m.geom_type[g] == GEOM_PLANE) {
      const F3 pn = colf(GM + 9 * g, 2);
      BX[12 * g] = dotf(pn, colf(cm, 0)); BX[12 * g + 1] = dotf(pn, colf(cm, 1)); BX[12 * g + 2] = dotf(pn, colf(cm, 2));
    }
  }
  barrier();
  const F3 origin = f3(0, 0, 0);
  // (tan(fovy / 2) per camera and cos(cutoff) per light come from the host, mjrl_create: in double precision on the
  // device they were 600 of a wave's 5 k instructions)
  const float t = consts[cam], aspect = (float)width / (float)height;
  const float inv_w = 2.0f / (float)width, inv_h = 2.0f / (float)height;
  // TWO RAYS PER LANE (round 4).  The wave's 128 rays cover a block of 16 x 8 pixels at a time, lane L the pixels
  // (r0 + L / 8, c0 + L % 8) and the one 8 columns to its right: a pair's arithmetic runs in packed fp32 (mjrl_rayf.h,
  // "two rays per lane"), and what a block costs per candidate geom whatever the rays do -- five v_readlane for its
  // record, the ballots, the loop -- is paid once for 128 rays instead of once for 64.  Lane g first tests geom g's
  // bounding sphere against the block's bounding cone (all geoms at once); the rays then visit only the geoms that passed,
  // each ray with its own bounding-sphere test before the type-specific one.  Both tests are conservative.
  const int bw = (width + 15) / 16;
  auto pixel_ray = [&](float px, float py) {        // px, py in pixel units, pixel centres at +0.5
    return normalizedf(mulf(cm, f3((px * inv_w - 1.0f) * t * aspect, (py * inv_h - 1.0f) * t, -1.0f)));
  };
  auto pixel_rays = [&](f2 px, float py) {
    return normalized2(mul2(cm, p3((px * splat2(inv_w) - splat2(1.0f)) * splat2(t) * splat2(aspect), splat2((py * inv_h - 1.0f) * t),
                                   splat2(-1.0f))));
  };
  // The bounding cones of the wave's blocks -- axis through the block's centre, half-angle from its corner rays: five
  // rays per block -- are computed 64 blocks at a time, lane j the cone of the j-th block to come, and parked in LDS
  // (five floats per block; held in registers they were five of the 128 the kernel may use, for all of it).
  float* CN = GK + 2 * m.ngeom;
  int ahead = 0;
  auto lane_f = [&](float v, int src) { return __int_as_float(wv::lane_int(__float_as_int(v), src)); };
  for (int blk = blk0; blk < blk1; blk += blk_step, ahead++) {
    if ((ahead & 63) == 0) {
      const int mine = blk + L * blk_step, mr0 = (mine / bw) * 8, mc0 = (mine % bw) * 16;
      const F3 cone_axis = pixel_ray(mc0 + 8.0f, mr0 + 4.0f);
      float cosmin = 1.0f;
      for (int k = 0; k < 4; k++)
        cosmin = fminf(cosmin, dotf(cone_axis, pixel_ray(mc0 + ((k & 1) ? 15.5f : 0.5f), mr0 + ((k & 2) ? 7.5f : 0.5f))));
      const float cone_cos = cosmin * (1.0f - 1e-5f) - 1e-6f;
      barrier();                         // (the cones of the 64 blocks before have been read)
      CN[5 * L] = cone_axis.x; CN[5 * L + 1] = cone_axis.y; CN[5 * L + 2] = cone_axis.z;
      CN[5 * L + 3] = cone_cos; CN[5 * L + 4] = fsqrt(fmaxf(1.0f - cone_cos * cone_cos, 0.0f));
      barrier();
    }
    const int r0 = (blk / bw) * 8, c0 = (blk % bw) * 16, from = ahead & 63;
    const F3 axis = ldf3(CN + 5 * from);
    const float cos_t = CN[5 * from + 3], sin_t = CN[5 * from + 4];
    const int r = r0 + (L >> 3), c = c0 + (L & 7);
    B2 inside; inside.x = r < height && c < width; inside.y = r < height && c + 8 < width;
    const P3 vec = pixel_rays(mk2(c + 0.5f, c + 8.5f), r + 0.5f);
    const P3 at_eye = splat3(origin);
    f2 best = splat2(-1.0f);
    int hit0 = -1, hit1 = -1;
    // the geoms in chunks of 64, lane g of a chunk standing for geom base + g (one chunk unless the level has more than
    // 64 geoms; the first chunk's records live in registers for the whole kernel, a later chunk's are fetched per block)
    // the block's edges per unit depth (pixel EDGES, so that every pixel centre's ray lies inside)
    const float e_a = (c0 * inv_w - 1.0f) * t * aspect, e_b = ((c0 + 16) * inv_w - 1.0f) * t * aspect;
    const float e_c = (r0 * inv_h - 1.0f) * t, e_d = ((r0 + 8) * inv_h - 1.0f) * t;
    auto visit = [&](int base, bool geom_on, int type_v, float rb_v, F3 rel_v) {
      bool cand = geom_on;
      if (geom_on && type_v != GEOM_PLANE) {
        const float along = dotf(rel_v, axis), perp = fsqrt(fmaxf(dotf(rel_v, rel_v) - along * along, 0.0f));
        // (perp cos - along sin is a lower bound of the centre's distance to the cone, negative inside it)
        cand = !(along + rb_v < 0.0f) && perp * cos_t - along * sin_t <= rb_v + 1e-5f * (1.0f + perp);
      }
      if (cand && type_v == GEOM_PLANE && !loose_culls) {
        // n . (lx X + ly Y - Z) is linear in (lx, ly): if it is >= 0 at the block's four corners no ray of the block
        // travels against the plane's normal (ray_geom2: denom > -1e-15 misses)
        const float* B = BX + 12 * (base + L);
        const float nx = B[0], ny = B[1], nz = B[2];
        const float d0 = nx * e_a + ny * e_c - nz, d1 = nx * e_b + ny * e_c - nz, d2 = nx * e_a + ny * e_d - nz, d3 = nx * e_b + ny * e_d - nz;
        cand = fminf(fminf(d0, d1), fminf(d2, d3)) < 1e-6f * (fabsf(nx) + fabsf(ny) + fabsf(nz));
      }
      if (cand && type_v == GEOM_BOX && !loose_culls) {
        const float* B = BX + 12 * (base + L);
        const float cx = B[0], x1 = B[1], x2 = B[2], x3 = B[3], cy = B[4], y1 = B[5], y2 = B[6], y3 = B[7];
        const float cz = B[8], z1 = B[9], z2 = B[10], z3 = B[11];
        auto behind = [&](float cw, float w1, float w2, float w3, float edge, float sign) {
          // signed distance of the box's nearest-to-inside corner from the plane sign (W + edge Z): centre + extent
          const float centre = sign * (cw + edge * cz);
          const float extent = fabsf(w1 + edge * z1) + fabsf(w2 + edge * z2) + fabsf(w3 + edge * z3);
          return centre + extent < -1e-4f * (1.0f + fabsf(centre) + extent);
        };
        cand = !(behind(cx, x1, x2, x3, e_a, 1.0f) || behind(cx, x1, x2, x3, e_b, -1.0f) ||
                 behind(cy, y1, y2, y3, e_c, 1.0f) || behind(cy, y1, y2, y3, e_d, -1.0f));
      }
      unsigned long long todo = wv::ballot(cand);
      while (todo) {
        const int g = __builtin_ctzll(todo);
        todo &= todo - 1;
        // (geom g's type, bounding radius and position come from lane g's registers: no memory latency per candidate)
        const int gt = wv::lane_int(type_v, g);
        const F3 rel = f3(lane_f(rel_v.x, g), lane_f(rel_v.y, g), lane_f(rel_v.z, g));
        B2 on = inside;
        // (a box has passed the block's frustum test: most of the block's rays do hit it, and its bounding sphere -- a
        // wall's is 10 m -- decides nothing the slab test does not decide at the same price)
        if (gt != GEOM_PLANE && (gt != GEOM_BOX || loose_culls)) {
          const float rb = lane_f(rb_v, g), rr = dotf(rel, rel);
          const f2 along = dot2(splat3(rel), vec), d2 = splat2(rr) - along * along;
          const float lim = rb * rb + 1e-5f * (1.0f + rr);
          on = and2(on, not2(or2(gt2(d2, splat2(lim)), lt2(along + splat2(rb), splat2(0.0f)))));
        }
        if (!wv::ballot(on.x || on.y)) continue;           // no ray of the wave comes near this geom
        const f2 x = ray_geom2(gt, on, rel, GM + 9 * (base + g), ldf3(GS + 3 * (base + g)), at_eye, vec);
        if (x.x >= 0 && (best.x < 0 || x.x < best.x)) { best.x = x.x; hit0 = base + g; }
        if (x.y >= 0 && (best.y < 0 || x.y < best.y)) { best.y = x.y; hit1 = base + g; }
      }
    };
    for (int base = 0; base < m.ngeom; base += 64) {       // (one chunk unless the level has more than 64 geoms)
      const int g = base + L < m.ngeom ? base + L : 0;
      const int type_g = base + L < m.ngeom ? (int)GK[2 * g] : -1;
      visit(base, type_g >= 0, type_g, GK[2 * g + 1], ldf3(GP + 3 * g));
    }
    unsigned packed0 = 0u, packed1 = 0u;
    // (the shading runs for the whole wave, rays without a hit computing on geom 0 and storing nothing: the shadow pass
    // below needs every lane in its role as a GEOM, whatever its pixels saw)
    if (wv::ballot(hit0 >= 0 || hit1 >= 0)) {
      // OpenGL's fixed-function lighting equation with the parameters MuJoCo documents (oracle/ora_step.c ora_shade, the
      // same arithmetic in double): emission + per light att * spot * (ambient + max(n.L, 0) diffuse + (n.H)^shininess
      // specular), the geom's rgba as ambient and diffuse material colour, clamped once at the end
      B2 shaded; shaded.x = hit0 >= 0; shaded.y = hit1 >= 0;
      const int hs0 = shaded.x ? hit0 : 0, hs1 = shaded.y ? hit1 : 0;
      const P3 P = vec * sel2(shaded, best, splat2(1.0f));
      const P3 n = pair3(geom_normalf(m.geom_type[hs0], ldf3(GP + 3 * hs0), GM + 9 * hs0, ldf3(GS + 3 * hs0), first3(P)),
                         geom_normalf(m.geom_type[hs1], ldf3(GP + 3 * hs1), GM + 9 * hs1, ldf3(GS + 3 * hs1), second3(P)));
      // the direction and distance of light li from the pair's surface points, and whether they lie in its cone at all
      auto light_geometry = [&](const float* T, bool directional, P3& Ld, f2& light_dist, f2& d2raw, f2& cc) {
        const F3 dir = ldf3(T + 3);
        Ld = splat3(dir * -1.0f);
        light_dist = splat2(3.0e38f); d2raw = splat2(1.0f); cc = splat2(1.0f);
        if (!directional) {
          const P3 to_light = splat3(ldf3(T)) - P;
          d2raw = dot2(to_light, to_light);
          const f2 d2 = fmax2(d2raw, splat2(1e-30f));          // (a point AT the light: its terms come out finite, then 0)
          const f2 inv = frsq2(d2);
          light_dist = d2 * inv;
          Ld = to_light * inv;
          cc = -dot2(Ld, splat3(dir));
        }
      };
      // The lights' shadows FIRST (MuJoCo draws them for every light with castshadow; oracle/ora_step.c ora_shade), as one
      // bit per light and ray, before the shading proper loads its materials and colours: a ray from the surface point
      // towards the light, and if another opaque geom lies on it the light's diffuse and specular terms go.  Which geoms can
      // lie on ANY of the block's shadow rays is decided once per block and light, lane g for geom g: the block's lit
      // points lie in a ball B (on the block's cone between the nearest and the farthest hit), their rays in the hull of
      // B and the light, and a geom whose bounding sphere stays clear of that hull is no candidate.
      unsigned dark0 = 0u, dark1 = 0u;
#pragma unroll 1
      for (int li = 1; li <= m.nlight && li < 32; li++) {
        const float* T = LT + LIGHT_FLOATS * li;
        if (wv::first_int(__float_as_int(T[21])) == 0) continue;
        const bool directional = wv::first_int(__float_as_int(T[20])) != 0;
        P3 Ld; f2 light_dist, d2raw, cc;
        light_geometry(T, directional, Ld, light_dist, d2raw, cc);
        B2 lit = and2(shaded, gt2(dot2(n, Ld), splat2(0.0f)));
        if (!directional) {
          lit = and2(lit, not2(lt2(d2raw, splat2(1e-30f))));
          if (T[9] > -1.5f) lit = and2(lit, not2(lt2(cc, splat2(T[9]))));
        }
        if (!wv::ballot(lit.x || lit.y)) continue;
        const B2 was_lit = lit;
        const F3 dir = ldf3(T + 3);
        const float tmin = wave_minf(fminf(lit.x ? best.x : 3.0e38f, lit.y ? best.y : 3.0e38f));
        const float tmax = wave_maxf(fmaxf(lit.x ? best.x : 0.0f, lit.y ? best.y : 0.0f));
        const F3 bc = axis * (0.5f * (tmin + tmax));
        const float br = 0.5f * (tmax - tmin) + tmax * sin_t * frcp(fmaxf(cos_t, 0.1f)) + 1e-4f * (1.0f + tmax);
        F3 toward = dir * -1.0f;
        float reach = 3.0e38f;
        if (!directional) {
          const F3 e = ldf3(T) - bc;
          const float e2 = fmaxf(dotf(e, e), 1e-30f), inv = rsqrtf(e2);
          toward = e * inv;
          reach = e2 * inv;
        }
        auto shadow_pass = [&](int base, bool geom_on, int type_v, float rb_v, F3 rel_v) {
          bool cand = geom_on;
          if (geom_on) {
            if (type_v == GEOM_PLANE) {
              // a ray reaches a plane's front only travelling against its normal; a positional light must lie behind it
              const F3 pn = colf(GM + 9 * (base + L), 2);
              cand = directional ? dotf(toward, pn) < 0.0f : dotf(ldf3(T) - rel_v, pn) < br;
            } else {
              const F3 w = rel_v - bc;
              const float along = fminf(fmaxf(dotf(w, toward), 0.0f), reach);
              const F3 off = w - toward * along;
              const float lim = rb_v + br;
              cand = dotf(off, off) <= lim * lim + 1e-5f * (1.0f + dotf(w, w));
              if (cand && type_v == GEOM_BOX && !loose_culls) {
                // (a wall's bounding sphere reaches every hull: on each of the box's own axes the hull -- the segment
                // from the ball's centre towards the light, fattened by the ball's radius -- must overlap the box's slab)
                const float* gm = GM + 9 * (base + L);
                const F3 p = mulTf(gm, bc - rel_v), dl = mulTf(gm, toward), sz = ldf3(GS + 3 * (base + L));
                const float len = directional ? 1.0e30f : reach, pad = br + 1e-4f * (1.0f + br);
                auto clear = [&](float pi, float di, float si) {
                  const float qi = pi + di * len;
                  return fminf(pi, qi) - pad > si || fmaxf(pi, qi) + pad < -si;
                };
                cand = !(clear(p.x, dl.x, sz.x) || clear(p.y, dl.y, sz.y) || clear(p.z, dl.z, sz.z));
              }
            }
          }
          unsigned long long todo = wv::ballot(cand);
          while (todo) {
            const int g = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int gt = wv::lane_int(type_v, g);
            const F3 rel = f3(lane_f(rel_v.x, g), lane_f(rel_v.y, g), lane_f(rel_v.z, g));
            // (the geom a point lies on is convex: it hides the light only where n.L <= 0)
            B2 on = lit;
            on.x = on.x && base + g != hs0; on.y = on.y && base + g != hs1;
            if (gt != GEOM_PLANE) {
              const float rb = lane_f(rb_v, g);
              const P3 w = splat3(rel) - P;
              const f2 ww = dot2(w, w), along = dot2(w, Ld), d2 = ww - along * along;
              const f2 lim = splat2(rb * rb) + splat2(1e-5f) * (splat2(1.0f) + ww);
              on = and2(on, not2(or2(or2(gt2(d2, lim), lt2(along + splat2(rb), splat2(0.0f))), gt2(along - splat2(rb), light_dist))));
            }
            if (!wv::ballot(on.x || on.y)) continue;
            const f2 x = ray_geom2(gt, on, rel, GM + 9 * (base + g), ldf3(GS + 3 * (base + g)), P, Ld);
            lit = and2(lit, not2(and2(ge2(x, splat2(0.0f)), lt2(x, light_dist))));
          }
        };
        for (int base = 0; base < m.ngeom; base += 64) {
          const int g = base + L < m.ngeom ? base + L : 0;
          const int type_g = base + L < m.ngeom ? (int)GK[2 * g] : -1;
          shadow_pass(base, type_g >= 0, type_g, GK[2 * g + 1], ldf3(GP + 3 * g));
        }
        if (was_lit.x && !lit.x) dark0 |= 1u << li;
        if (was_lit.y && !lit.y) dark1 |= 1u << li;
      }
      const f2 spec_m = mk2(MP[3 * hs0], MP[3 * hs1]), shin = mk2(MP[3 * hs0 + 1], MP[3 * hs1 + 1]);
      const f2 emis = mk2(MP[3 * hs0 + 2], MP[3 * hs1 + 2]);
      const P3 mat = pair3(ldf3(GC + 3 * hs0), ldf3(GC + 3 * hs1));
      const P3 V = splat3(f3(-LT[3], -LT[4], -LT[5]));        // towards the viewer (at infinity): the camera's +z axis
      P3 col = p3(emis * mat.x, emis * mat.y, emis * mat.z);
      auto pow2 = [&](f2 x, f2 y) { return mk2(fast_pow(x.x, y.x), fast_pow(x.y, y.y)); };
#pragma unroll 1
      for (int li = 0; li <= m.nlight; li++) {
        const float* T = LT + LIGHT_FLOATS * li;
        const bool directional = wv::first_int(__float_as_int(T[20])) != 0;
        P3 Ld; f2 light_dist, d2raw, cc;
        light_geometry(T, directional, Ld, light_dist, d2raw, cc);
        f2 scale = splat2(1.0f);
        if (!directional) {
          const f2 d2 = fmax2(d2raw, splat2(1e-30f));
          scale = frcp2(splat2(T[6]) + splat2(T[7]) * light_dist + splat2(T[8]) * d2);
          if (T[9] > -1.5f) scale = sel2(lt2(cc, splat2(T[9])), splat2(0.0f), scale * pow2(fmax2(cc, splat2(0.0f)), splat2(T[10])));
          scale = sel2(lt2(d2raw, splat2(1e-30f)), splat2(0.0f), scale);
        }
        f2 nl = dot2(n, Ld);
        B2 facing = gt2(nl, splat2(0.0f));
        if (li < 32) { facing.x = facing.x && !((dark0 >> li) & 1u); facing.y = facing.y && !((dark1 >> li) & 1u); }
        nl = sel2(facing, nl, splat2(0.0f));
        const P3 H = Ld + V;
        const f2 hh = dot2(H, H);
        const f2 nh = sel2(gt2(hh, splat2(1e-30f)), dot2(n, H) * frsq2(fmax2(hh, splat2(1e-30f))), splat2(0.0f));
        const f2 sp = sel2(and2(facing, gt2(nh, splat2(0.0f))), pow2(fmax2(nh, splat2(0.0f)), shin) * spec_m, splat2(0.0f));
        col.x = col.x + scale * (splat2(T[11]) * mat.x + nl * splat2(T[14]) * mat.x + sp * splat2(T[17]));
        col.y = col.y + scale * (splat2(T[12]) * mat.y + nl * splat2(T[15]) * mat.y + sp * splat2(T[18]));
        col.z = col.z + scale * (splat2(T[13]) * mat.z + nl * splat2(T[16]) * mat.z + sp * splat2(T[19]));
      }
      auto level = [&](float v) { return (unsigned)(unsigned char)(255.0f * fminf(fmaxf(v, 0.0f), 1.0f) + 0.5f); };
      if (shaded.x) packed0 = level(col.x.x) | (level(col.y.x) << 8) | (level(col.z.x) << 16);
      if (shaded.y) packed1 = level(col.x.y) | (level(col.y.y) << 8) | (level(col.z.y) << 16);
    }
    // A half block's 8 rows of 8 pixels are 8 x 24 bytes: written as 48 dwords, 6 per row, each put together from two
    // neighbouring lanes' pixels (round 2 stored 3 single bytes per lane: 192 byte stores per block, and four times the
    // pixels' bytes in HBM write traffic, profiles/r02_pmc_render_fp64.txt).  Needs dword-aligned rows, i.e. a width
    // that is a multiple of 4, and a half block wholly inside the image; anything else keeps the byte stores.
    auto store_half = [&](unsigned packed, int ch0, bool in_image) {
      const bool whole = (row_bytes & 3) == 0 && r0 + 8 <= height && ch0 + 8 <= width;
      if (whole) {
        const int prow = L / 6, j = L - 6 * prow;             // lanes 0..47: dword j of the block's pixel row prow
        const int b = 4 * j, pa = b / 3, o = b - 3 * pa;       // its first byte belongs to pixel pa of that row, byte o
        const int src = (prow & 7) * 8 + pa;
        const unsigned lo = (unsigned)wv::shfl((int)packed, src), hi = (unsigned)wv::shfl((int)packed, (src + 1) & 63);
        const unsigned long long q = (unsigned long long)lo | ((unsigned long long)hi << 24);
        if (L < 48) *(unsigned*)(img + (size_t)(r0 + prow) * row_bytes + 3 * ch0 + 4 * j) = (unsigned)(q >> (8 * o));
      } else if (in_image) {
        unsigned char* at = img + (size_t)r * row_bytes + 3 * (ch0 + (L & 7));
        at[0] = (unsigned char)packed; at[1] = (unsigned char)(packed >> 8); at[2] = (unsigned char)(packed >> 16);
      }
    };
    store_half(packed0, c0, inside.x);
    if (c0 + 8 < width) store_half(packed1, c0 + 8, inside.y);
  }
}

// Four waves per SIMD.  The pair kernel wanted 137 registers with the geoms' cull records and the blocks' cones held in
// registers (held to 128 it spilled 52 B per lane); with both read from LDS per block it takes 111 and no scratch.  Five
// waves (96 registers) spill 56 B into the candidate loop: 84 us against 79.  tools/render_probe.sh.
#ifndef MJRL_RENDER_WAVES
#define MJRL_RENDER_WAVES 4
#endif
#define MJRL_RENDER_OCC __attribute__((amdgpu_waves_per_eu(MJRL_RENDER_WAVES, MJRL_RENDER_WAVES)))
__global__ __launch_bounds__(64) MJRL_RENDER_OCC void mjrl_render_kernel(DevModel m, const double* scene, int n_env, int width, int height,
                                                         int tiles, unsigned char* rgb, const int* variant,
                                                         const double* variant_rgba, const float* consts, int loose_culls) {
  extern __shared__ float ldsf[];
  const int env = blockIdx.x, cam = blockIdx.y / tiles, tile = blockIdx.y % tiles;
  const int nblock = ((width + 15) / 16) * ((height + 7) / 8);          // blocks of 16 x 8 pixels: two rays per lane
  const int blk0 = (int)((long long)tile * nblock / tiles), blk1 = (int)((long long)(tile + 1) * nblock / tiles);
  const double* rgba_tab = (variant && variant_rgba) ? variant_rgba + (size_t)variant[env] * 4 * m.ngeom : nullptr;
  render_body(m, scene + (size_t)env * scene_doubles(m), cam, width, height, blk0, blk1, 1, ldsf,
              rgb + ((size_t)env * m.ncam + cam) * width * height * 3, 3 * width, rgba_tab, consts, loose_culls);
}

// the ray kernel's LDS: geom positions, matrices and sizes
inline size_t render_lds_bytes(const DevModel& m) {
  // geom positions, matrices, sizes, material properties, colours | lights | camera matrix (padded to 12) | boxes' frustum table
  // | per geom: type (-1: transparent, never a candidate) and bounding radius | the cones of 64 blocks
  return (21 * (size_t)m.ngeom + LIGHT_FLOATS * ((size_t)m.nlight + 1) + 12 + 12 * (size_t)m.ngeom + 2 * (size_t)m.ngeom + 5 * 64) * sizeof(float);
}

std::string g_create_error;

}  // namespace

struct mjrl_env {
  std::vector<char> h_blob;
  DevModel hm{}, dm{};
  mj::Lay lay{};
  hipModule_t spec_module = nullptr;     // model-specialised step kernel, if one was attached (mjrl_load_kernel)
  hipFunction_t spec_fn = nullptr;
  bool spec_diag = false;                // that code object was built with the diagnostics (-DMJRL_DIAG)
  bool big = false;                      // the generic kernels of the roomy kind serve this batch (mj::pgs_roomy, or few)
  bool few = false;                      // the batch leaves every SIMD at most one wave (StepArgs::few)
  void* d_blob = nullptr;
  DevModel* d_model = nullptr;     // device copy of `dm`
  int32_t* d_lane_rec = nullptr;   // the lanes' records of the model (mj::build_lane_records), StepArgs::lane_rec
  int n_env = 0, device = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;
  double *qpos = nullptr, *qvel = nullptr, *ctrl = nullptr, *warm = nullptr, *sens = nullptr, *dbg = nullptr;
  unsigned long long* overflow = nullptr;   // [2] sticky cap-overflow counters (mjrl_cap_overflows)
  double* scene = nullptr;         // [n_env][scene_doubles] geom, camera and light frames for the ray caster
  double* reset_scene = nullptr;   // [scene_doubles] the same at the reset state
  float* render_consts = nullptr;  // [ncam + nlight] tan(fovy / 2) per camera, cos(cutoff) per light (-2: no cone)
  // mjrl_set_scene_cache: the step kernel leaves every copy's scene row as its forward pass computed it; scene_valid says
  // that every row belongs to the copies' last forward pass (no state has been written by hand since)
  bool scene_on = false, scene_valid = false;
  double* rk = nullptr;            // [n_env][nq + 3 nv] Runge-Kutta scratch (models with <option integrator="RK4">)
  int* stats = nullptr;            // [n_env][4] ncon, nefc, solver sweeps, warning bits of each copy's last physics frame
  int* timestep = nullptr;
  unsigned char* d_mask = nullptr;
  // reset image: what mj_forward leaves at the reset state (the same for every copy), computed once at create
  double *reset_warm = nullptr, *reset_sens = nullptr;
  // object tags (mjrl_set_tag_tables), global id of copy 0, per-copy level variants (mjrl_set_variants)
  int32_t *d_tag_adr = nullptr, *d_tag_num = nullptr, *d_tag_ref = nullptr;
  int n_tag = 0, env_base = 0;
  std::vector<int32_t> h_tag_num;
  int *variant = nullptr, *episode = nullptr, n_variant = 0;
  unsigned long long variant_seed = 0;
  double* variant_rgba = nullptr;         // [n_variant][ngeom][4]
  // camera encoder (mjrl_encoder_load): packed bf16 weight fragments, biases, scratch; camera latents in the observation
  void *enc_w1 = nullptr, *enc_w2 = nullptr, *enc_wd = nullptr;
  float *enc_b1 = nullptr, *enc_b2 = nullptr, *enc_bd = nullptr;
  unsigned short* enc_a2 = nullptr;        // [enc_cap][16384] conv output (bf16)
  float* enc_part = nullptr;               // [DENSE_KSPLIT][enc_cap][16 enc_tiles] partial tiles of the dense layer
  int enc_part_tiles = 0;
  unsigned char* enc_rgb = nullptr;        // [n_env][ncam][64][64][3] frames of the camera observation
  int* enc_obs_row = nullptr;              // [n_env * ncam] first latent slot of the image in the flat observation tensor, -1: none
  int enc_latent = 0, enc_tiles = 0, enc_relu = 1, enc_cap = 0, n_cam_obs = 0;
  std::vector<int32_t> h_agent_cam;
  const unsigned char* step_reset_mask = nullptr;   // caller-owned device mask of the in-launch reset (mjrl_set_step_reset_mask)
  unsigned char* auto_mask = nullptr;               // [n_env] "the copy's episode ended in its last step" (mjrl_set_autoreset)
  int auto_mode = 0;
  int io_agent = -1, obs_f32 = 0;   // mjrl_set_io_layout
  // tables
  int n_agent = 0, obs_dim = 0, scatter_mode = 0, max_steps = 1024;
  std::vector<int32_t> h_gather;                 // [n_agent][obs_dim]
  std::vector<std::vector<int32_t>> h_scatter;   // per agent
  int32_t *d_gather = nullptr, *d_scatter = nullptr;
  int scatter_act_dim = -1;
  // fused plugin program
  std::vector<int32_t> h_obs_len;                // physical observation length per agent
  int base_obs_dim = 0, n_extra = 0, n_op = 0, n_slot = 0;
  int32_t *d_prog_i = nullptr, *d_agent_body = nullptr, *d_obs_len = nullptr;
  double *d_prog_f = nullptr, *store = nullptr;
  // longest-first dispatch: two generations of work buckets (read the previous launch's, fill the next one's)
  // three generations of work buckets: a launch reads one, files into the next and clears the third for the launch
  // after it (a hipMemsetAsync per step was a 5 us kernel of its own, 2 % of the step)
  int* lpt_count[3] = {nullptr, nullptr, nullptr};
  unsigned* lpt_mask[3] = {nullptr, nullptr, nullptr};     // [LPT_BUCKETS][lpt_words] bit sets
  int lpt_words = 0;
  int lpt_cur = 0;
  bool lpt_valid = false, lpt_enabled = true;
  int stop_after = 0;              // diagnostic (mjrl_step_truncated)
  // forward-pass frames kept for host-side plugin queries
  double* frames = nullptr;
  bool frames_valid = false;
  // staging for the host-buffer entry points
  double *s_act = nullptr, *s_obs = nullptr, *s_rew = nullptr;
  unsigned char *s_term = nullptr, *s_trunc = nullptr;
  size_t s_act_n = 0, s_obs_n = 0;
  // pinned host buffers mapped into the device's address space (mjrl_host_buffers / mjrl_step_pinned)
  double *p_act = nullptr, *p_obs = nullptr, *p_rew = nullptr;
  unsigned char *p_term = nullptr, *p_trunc = nullptr;
  size_t p_act_n = 0, p_obs_n = 0, p_na = 0;     // elements allocated: actions, observations, rewards / flags
  void* p_dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // their device addresses, and the host buffers
  void* p_dev_of[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // they were looked up for
  std::string err;
};

#define MJRL_FAIL(env, code, ...)                              \
  do {                                                         \
    char buf_[512];                                            \
    snprintf(buf_, sizeof(buf_), __VA_ARGS__);                 \
    (env)->err = buf_;                                         \
    return (code);                                             \
  } while (0)

#define MJRL_HIP(env, call)                                                                    \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) MJRL_FAIL(env, 100 + (int)e_, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

// Every entry point works on the handle's device whatever the calling thread's current device is, and leaves the
// caller's current device as it found it.
struct DeviceGuard {
  int prev = -1, want;
  explicit DeviceGuard(int device) : want(device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != want) hipSetDevice(want);
  }
  ~DeviceGuard() { if (prev >= 0 && prev != want) hipSetDevice(prev); }
};
#define MJRL_ENTER(env)                  \
  if (!(env)) return 1;                  \
  DeviceGuard device_guard_((env)->device)

extern "C" {

#define MJRL_STR_(x) #x
#define MJRL_STR(x) MJRL_STR_(x)
const char* mjrl_version(void) { return "mjrl-hip 0.7 (blob layout " MJRL_STR(MJRL_BLOB_VERSION) ", gfx950)"; }

const char* mjrl_last_error(const mjrl_env* env) { return env ? env->err.c_str() : g_create_error.c_str(); }

void mjrl_destroy(mjrl_env* e) {
  if (!e) return;
  DeviceGuard guard(e->device);
  void* ptrs[] = {e->scene, e->rk, e->enc_w1, e->enc_w2, e->enc_wd, e->enc_b1, e->enc_b2, e->enc_bd, e->enc_a2, e->enc_part, e->enc_rgb, e->enc_obs_row, e->d_tag_adr, e->d_tag_num, e->d_tag_ref, e->variant, e->episode, e->variant_rgba, e->stats, e->reset_warm, e->reset_sens, e->d_blob, e->d_model, e->d_lane_rec, e->qpos, e->qvel, e->ctrl, e->warm, e->sens, e->dbg, e->timestep, e->d_mask, e->d_gather,
                  e->d_scatter, e->s_act, e->s_obs, e->s_rew, e->s_term, e->s_trunc, e->d_prog_i, e->d_agent_body,
                  e->d_obs_len, e->d_prog_f, e->store, e->frames, e->lpt_count[0], e->lpt_count[1], e->lpt_count[2],
                  e->lpt_mask[0], e->lpt_mask[1], e->lpt_mask[2], e->overflow, e->auto_mask, e->reset_scene, e->render_consts};
  for (void* p : ptrs) if (p) hipFree(p);
  void* pinned[] = {e->p_act, e->p_obs, e->p_rew, e->p_term, e->p_trunc};
  for (void* p : pinned) if (p) hipHostFree(p);
  if (e->spec_module) hipModuleUnload(e->spec_module);
  if (e->own_stream) hipStreamDestroy(e->own_stream);
  delete e;
}

static int launch_reset(mjrl_env* e, const unsigned char* d_mask, double* d_obs = nullptr) {
  if (d_obs && !e->d_gather) MJRL_FAIL(e, 3, "reset: observations requested but no gather table is set");
  hipLaunchKernelGGL(mjrl_reset_kernel, dim3(e->n_env), dim3(64), 0, e->stream, e->dm, e->qpos, e->qvel, e->ctrl, e->warm,
                     e->sens, e->timestep, d_mask, e->n_env, e->store, e->n_agent * e->n_slot, e->reset_warm, e->reset_sens,
                     e->d_gather, e->n_agent, e->obs_dim, d_obs, e->variant, e->episode, e->n_variant, e->variant_seed,
                     e->env_base, e->auto_mask, e->scene_on ? e->scene : nullptr, e->reset_scene, e->io_agent, e->obs_f32);
  MJRL_HIP(e, hipGetLastError());
  return 0;
}

static int launch_step(mjrl_env* e, const double* d_actions, int act_dim, int skip_frames, double* d_obs, double* d_reward,
                       uint8_t* d_term, uint8_t* d_trunc, double* d_dbg, int dbg_stage, int forward_only,
                       unsigned long long* d_stamps = nullptr, unsigned long long* d_timeline = nullptr);
static int launch_encoder(mjrl_env* e, const uint8_t* d_rgb, int n_img, float* d_latent, double* d_obs, const int* d_obs_row);

// argument values the self-check kernels compare (never dereferenced)
static mj::StepArgs selfcheck_args() {
  mj::StepArgs a{};
  a.qpos = (double*)0x1000; a.timestep = (int*)0x2000; a.obs = (double*)0x3000; a.n_env = 1234567; a.max_steps = 7654321;
  a.stats = (int*)0x4000; a.few = 91; a.lpt_words = 4242; a.overflow = (unsigned long long*)0x5000; a.stop_after = 77;
  return a;
}

int mjrl_create(const void* blob, size_t nbytes, int n_env, int device_id, unsigned flags, mjrl_env** out) {
  if (!blob || !out || n_env <= 0) { g_create_error = "mjrl_create: bad arguments"; return 1; }
  mjrl_env* e = new mjrl_env();
  auto fail = [&](int code, const std::string& msg) { g_create_error = msg; mjrl_destroy(e); return code; };
  e->h_blob.assign((const char*)blob, (const char*)blob + nbytes);
  int rc = mjrl_model_from_blob(&e->hm, e->h_blob.data(), nbytes, e->h_blob.data());
  if (rc) return fail(2, "mjrl_create: model blob rejected (magic/version/size), code " + std::to_string(rc));
  const DevModel& m = e->hm;
  if (m.nv < 1 || m.nv > 64 || m.nbody > 64 || m.njnt > 64 || m.ngeom > 128)
    return fail(3, "mjrl_create: the model must fit one wavefront (1 <= nv <= 64; nbody, njnt <= 64 after the compiler has "
                   "folded the bodies that cannot move; ngeom <= 128): the dof- and body-indexed stages have no second pass");
  if (m.maxdofdepth + 1 > mj::MAX_DOF_DEPTH) return fail(3, "mjrl_create: kinematic chains deeper than 8 dofs are not supported");
  if (m.nconmax > 64 || m.nconmax < 1) return fail(3, "mjrl_create: nconmax must be in 1..64");
  if (m.njmax > 511 || m.njmax < 1) return fail(3, "mjrl_create: njmax must be in 1..511");
  if (m.pair_kmax != 1 && m.pair_kmax != 2 && m.pair_kmax != 4 && m.pair_kmax != 8 && m.pair_kmax != 16)
    return fail(3, "mjrl_create: bad pair_kmax");
  if (m.integrator != 0 && m.integrator != 1) return fail(3, "mjrl_create: unknown integrator (0 Euler, 1 RK4)");
  mj::make_layout(m, e->lay);
  size_t lds_bytes = (size_t)e->lay.total * sizeof(double);
  if (lds_bytes > 160 * 1024) return fail(3, "mjrl_create: env working set exceeds the 160 KiB LDS of a CU");
  e->n_env = n_env;
  e->device = device_id;
  int caller_device = -1;
  hipGetDevice(&caller_device);
  struct Restore { int d; ~Restore() { if (d >= 0) hipSetDevice(d); } } restore{caller_device};
  hipError_t he = hipSetDevice(device_id);
  if (he != hipSuccess) return fail(4, std::string("hipSetDevice: ") + hipGetErrorString(he));
#define CK(call) do { he = (call); if (he != hipSuccess) return fail(5, std::string(#call ": ") + hipGetErrorString(he)); } while (0)
  CK(hipStreamCreate(&e->own_stream));
  e->stream = e->own_stream;
  CK(hipMalloc(&e->d_blob, nbytes));
  CK(hipMemcpy(e->d_blob, blob, nbytes, hipMemcpyHostToDevice));
  mjrl_model_from_blob(&e->dm, e->h_blob.data(), nbytes, e->d_blob);
  CK(hipMalloc(&e->d_model, sizeof(DevModel)));
  CK(hipMemcpy(e->d_model, &e->dm, sizeof(DevModel), hipMemcpyHostToDevice));
  {
    // what a lane reads from the model for itself, as records (mjrl_step.h, lane records): the loader functions of the
    // kernel run here, once, on the host copy of the model
    std::vector<int32_t> rec((size_t)mj::LANE_REC_INTS, 0);
    mj::build_lane_records(e->hm, e->lay, rec.data());
    CK(hipMalloc(&e->d_lane_rec, sizeof(int32_t) * rec.size()));
    CK(hipMemcpy(e->d_lane_rec, rec.data(), sizeof(int32_t) * rec.size(), hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&e->qpos, sizeof(double) * n_env * m.nq));
  CK(hipMalloc(&e->qvel, sizeof(double) * n_env * m.nv));
  CK(hipMalloc(&e->ctrl, sizeof(double) * n_env * (m.nu > 0 ? m.nu : 1)));
  CK(hipMalloc(&e->warm, sizeof(double) * n_env * m.nv));
  if (m.integrator == 1) CK(hipMalloc(&e->rk, sizeof(double) * (size_t)n_env * (m.nq + 3 * m.nv)));
  CK(hipMalloc(&e->stats, sizeof(int) * 4 * (size_t)n_env));
  CK(hipMemset(e->stats, 0, sizeof(int) * 4 * (size_t)n_env));
  CK(hipMalloc(&e->overflow, sizeof(unsigned long long) * 3));
  CK(hipMemset(e->overflow, 0, sizeof(unsigned long long) * 3));
  CK(hipMalloc(&e->sens, sizeof(double) * n_env * (m.nsensordata > 0 ? m.nsensordata : 1)));
  CK(hipMalloc(&e->timestep, sizeof(int) * n_env));
  CK(hipMalloc(&e->episode, sizeof(int) * n_env));          // resets so far, per copy (kept whether or not variants are on)
  CK(hipMemset(e->episode, 0, sizeof(int) * n_env));
  CK(hipMalloc(&e->d_mask, n_env));
  e->lpt_enabled = !(flags & 1u);
  e->lpt_words = (n_env + 31) / 32;
  for (int g = 0; g < 3; g++) {
    CK(hipMalloc(&e->lpt_count[g], sizeof(int) * mj::LPT_BUCKETS));
    CK(hipMemset(e->lpt_count[g], 0, sizeof(int) * mj::LPT_BUCKETS));
    CK(hipMalloc(&e->lpt_mask[g], sizeof(unsigned) * mj::LPT_BUCKETS * (size_t)e->lpt_words));
    CK(hipMemset(e->lpt_mask[g], 0, sizeof(unsigned) * mj::LPT_BUCKETS * (size_t)e->lpt_words));
  }
  {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, device_id));
    // (the rule assumes that the handle has the device to itself; a caller that steps several handles side by side --
    // train and eval envs, half-batches on two streams -- says so with flags bit 1, bit 2 forces the other way)
    e->few = n_env <= 4 * prop.multiProcessorCount;
    if (flags & 2u) e->few = false;
    if (flags & 4u) e->few = true;
    if (const char* f = getenv("MJRL_FEW")) e->few = atoi(f) != 0;      // (tests: either kind of solver forms at any batch size)
  }
  {
    // the production kernels read their arguments at a fixed offset of the kernarg segment (mjrl_step.h,
    // kernarg_step_args): checked once per process with a kernel of the same signature
    static bool kernarg_checked = false;
    if (!kernarg_checked) {
      int* d_ok = nullptr;
      int ok = 0;
      CK(hipMalloc(&d_ok, sizeof(int)));
      CK(hipMemset(d_ok, 0, sizeof(int)));
      hipLaunchKernelGGL(mjrl_selfcheck_kernel, dim3(1), dim3(64), 0, e->stream, e->d_model, selfcheck_args(), d_ok);
      CK(hipGetLastError());
      CK(hipMemcpyAsync(&ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, e->stream));
      CK(hipStreamSynchronize(e->stream));
      hipFree(d_ok);
      if (!ok) return fail(6, "mjrl_create: the step kernels' arguments are not where the kernels read them "
                              "(kernarg offset self-check failed): this build of libmjrl_hip.so must not step anything");
      kernarg_checked = true;
    }
  }
  e->big = e->few || mj::pgs_roomy(e->hm, e->lay);
  // (a batch whose waves are all resident at once has no dispatch order to improve: its copies are stepped by workgroup
  // id, without the look-up of the longest-first tables in every wave's prologue and without the filing at its end)
  if (e->few) e->lpt_enabled = false;
  CK(hipFuncSetAttribute((const void*)mjrl_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CK(hipFuncSetAttribute((const void*)mjrl_step_kernel_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CK(hipFuncSetAttribute((const void*)mjrl_step_kernel_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CK(hipFuncSetAttribute((const void*)mjrl_step_kernel_big_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CK(hipFuncSetAttribute((const void*)mjrl_camera_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
#undef CK
  // the reset image: reset every copy (zero warm start, as mj_resetData leaves it), run mj_forward once, keep copy 0's
  // warm start and sensor readings
  if (launch_reset(e, nullptr)) return fail(6, e->err);
  if (launch_step(e, nullptr, 0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1)) return fail(6, e->err);
  double *w0 = nullptr, *s0 = nullptr;
  he = hipMalloc(&w0, sizeof(double) * m.nv);
  if (he == hipSuccess) he = hipMalloc(&s0, sizeof(double) * (m.nsensordata > 0 ? m.nsensordata : 1));
  if (he == hipSuccess) he = hipMemcpyAsync(w0, e->warm, sizeof(double) * m.nv, hipMemcpyDeviceToDevice, e->stream);
  if (he == hipSuccess && m.nsensordata > 0)
    he = hipMemcpyAsync(s0, e->sens, sizeof(double) * m.nsensordata, hipMemcpyDeviceToDevice, e->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
  e->reset_warm = w0; e->reset_sens = s0;
  if (he == hipSuccess) he = hipMemset(e->episode, 0, sizeof(int) * n_env);     // (the reset above was not an episode's)
  if (he == hipSuccess && m.ncam > 0) {         // the ray caster's constants, and its scene row at the reset state
    std::vector<float> rc(m.ncam + m.nlight);
    const double pi = 3.14159265358979323846;
    for (int c = 0; c < m.ncam; c++) rc[c] = (float)tan(0.5 * m.cam_fovy[c] * pi / 180.0);
    for (int li = 0; li < m.nlight; li++)         // (cutoff >= 180: no cone -- a cosine no (-L . dir) can fall below)
      rc[m.ncam + li] = m.light_cutoff[li] < 180.0 ? (float)cos(m.light_cutoff[li] * pi / 180.0) : -2.0f;
    he = hipMalloc(&e->render_consts, sizeof(float) * rc.size());
    if (he == hipSuccess) he = hipMemcpy(e->render_consts, rc.data(), sizeof(float) * rc.size(), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMalloc(&e->reset_scene, sizeof(double) * mj::scene_doubles(m));
    if (he == hipSuccess) {
      hipLaunchKernelGGL(mjrl_camera_frames_kernel, dim3(1), dim3(64), lds_bytes, e->stream, e->dm, e->dm.qpos0, 1, e->reset_scene);
      he = hipGetLastError();
    }
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
  }
  if (he != hipSuccess) return fail(6, std::string("reset image: ") + hipGetErrorString(he));
  *out = e;
  return 0;
}

int mjrl_set_stream(mjrl_env* e, void* hip_stream) {
  MJRL_ENTER(e);
  hipStream_t next = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
  if (next == e->stream) return 0;
  MJRL_HIP(e, hipStreamSynchronize(e->stream));   // work queued on the old stream finishes before the switch
  e->stream = next;
  return 0;
}

int mjrl_sync(mjrl_env* e) {
  MJRL_ENTER(e);
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  return 0;
}

int mjrl_set_max_steps(mjrl_env* e, int max_steps) {
  MJRL_ENTER(e);
  e->max_steps = max_steps; return 0; }

// device gather table = physical part [base_obs_dim] + n_extra slots per agent owned by the fused program (-2)
static int upload_camera_rows(mjrl_env* e);

static int upload_gather(mjrl_env* e) {
  int dim = e->base_obs_dim + e->n_extra + e->n_cam_obs;
  std::vector<int32_t> table((size_t)e->n_agent * dim, -1);
  for (int a = 0; a < e->n_agent; a++) {
    for (int k = 0; k < e->base_obs_dim; k++) table[(size_t)a * dim + k] = e->h_gather[(size_t)a * e->base_obs_dim + k];
    // slots the step kernel's gather leaves alone: the fused program's, then the camera latents (written by the encoder).
    // An agent without a camera has no image and nobody writes its latent slots: they stay -1, which the gather writes
    // as 0 every step (include/mjrl.h: "slots of an agent without a camera read 0").
    const bool has_cam = a < (int)e->h_agent_cam.size() && e->h_agent_cam[a] >= 0;
    for (int k = 0; k < e->n_extra + (has_cam ? e->n_cam_obs : 0); k++) table[(size_t)a * dim + e->h_obs_len[a] + k] = -2;
  }
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  if (e->d_gather) { hipFree(e->d_gather); e->d_gather = nullptr; }
  if (e->d_obs_len) { hipFree(e->d_obs_len); e->d_obs_len = nullptr; }
  MJRL_HIP(e, hipMalloc(&e->d_gather, sizeof(int32_t) * std::max<size_t>(table.size(), 1)));
  MJRL_HIP(e, hipMemcpy(e->d_gather, table.data(), sizeof(int32_t) * table.size(), hipMemcpyHostToDevice));
  MJRL_HIP(e, hipMalloc(&e->d_obs_len, sizeof(int32_t) * e->n_agent));
  MJRL_HIP(e, hipMemcpy(e->d_obs_len, e->h_obs_len.data(), sizeof(int32_t) * e->n_agent, hipMemcpyHostToDevice));
  e->obs_dim = dim;
  e->s_obs_n = 0;     // staging is re-sized on the next host-buffer step
  if (e->s_obs) { hipFree(e->s_obs); e->s_obs = nullptr; }
  return e->n_cam_obs ? upload_camera_rows(e) : 0;
}

// image (env, cam) -> index of its first latent slot in the flat observation tensor [n_env][n_agent][obs_dim]
static int upload_camera_rows(mjrl_env* e) {
  const int ncam = e->hm.ncam;
  std::vector<int> rows((size_t)e->n_env * ncam, -1);
  for (int env = 0; env < e->n_env; env++)
    for (int a = 0; a < e->n_agent && a < (int)e->h_agent_cam.size(); a++) {
      const int cam = e->h_agent_cam[a];
      if (cam >= 0) rows[(size_t)env * ncam + cam] = ((size_t)env * e->n_agent + a) * e->obs_dim + e->h_obs_len[a] + e->n_extra;
    }
  if (e->enc_obs_row) { hipFree(e->enc_obs_row); e->enc_obs_row = nullptr; }
  MJRL_HIP(e, hipMalloc(&e->enc_obs_row, sizeof(int) * rows.size()));
  MJRL_HIP(e, hipMemcpy(e->enc_obs_row, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice));
  return 0;
}

int mjrl_set_program(mjrl_env* e, int n_op, const int32_t* prog_i, const double* prog_f, int n_slot, int n_extra_obs,
                     const int32_t* agent_body) {
  MJRL_ENTER(e);
  if (!e->n_agent || e->h_obs_len.empty()) MJRL_FAIL(e, 1, "set_program: set the gather tables first");
  if (e->n_agent > mj::MAX_AGENT) MJRL_FAIL(e, 1, "set_program: at most %d agents", (int)mj::MAX_AGENT);
  if (n_op < 0 || n_slot < 0 || n_extra_obs < 0) MJRL_FAIL(e, 1, "set_program: negative size");
  auto pf_ok = [](double seed) { return seed >= 0 && seed < 9007199254740992.0 && seed == (double)(unsigned long long)seed; };
  for (int op = 0; op < n_op; op++) {
    const int32_t* pi = prog_i + 8 * op;
    bool ok = true;
    if (pi[0] == mj::OP_LANGUAGE) ok = pi[2] >= 0 && pi[2] < n_slot && pi[3] >= 0 && pi[3] < n_extra_obs && pi[1] >= 0 && e->n_agent >= 2;
    else if (pi[0] == mj::OP_DIST_REWARD || pi[0] == mj::OP_DIST_DONE) {
      if (pi[1] == 2) ok = pi[2] >= 0 && pi[2] < e->n_tag && e->h_tag_num[pi[2]] > 0 && pi[5] >= 0 && pi[5] < n_slot;
      else ok = (pi[1] == 0 || pi[1] == 1) && pi[2] >= 0 && pi[2] < (pi[1] == 0 ? e->hm.nbody : e->hm.ngeom);
      if (pi[0] == mj::OP_DIST_REWARD) ok = ok && pi[3] < n_slot && (pi[4] == 0 || pi[3] >= 0);
    } else if (pi[0] == mj::OP_TARGET) {
      ok = pi[1] >= 0 && pi[1] < e->n_tag && e->h_tag_num[pi[1]] > 0 && pi[2] >= 0 && pi[2] < n_slot && pi[3] < n_slot &&
           pi[5] < n_slot && pi[4] >= 0 && pi[4] + (pi[3] >= 0 ? 4 : 3) <= n_extra_obs && pf_ok(prog_f[4 * op + 2]);
    } else ok = false;
    if (!ok) MJRL_FAIL(e, 2, "set_program: op %d (kind %d) is malformed", op, pi[0]);
  }
  for (int a = 0; a < e->n_agent; a++)
    if (agent_body[a] < 0 || agent_body[a] >= e->hm.nbody) MJRL_FAIL(e, 2, "set_program: agent body id out of range");
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  for (void** p : {(void**)&e->d_prog_i, (void**)&e->d_prog_f, (void**)&e->d_agent_body, (void**)&e->store})
    if (*p) { hipFree(*p); *p = nullptr; }
  e->n_op = n_op; e->n_slot = n_slot; e->n_extra = n_extra_obs;
  MJRL_HIP(e, hipMalloc(&e->d_prog_i, sizeof(int32_t) * 8 * std::max(n_op, 1)));
  MJRL_HIP(e, hipMalloc(&e->d_prog_f, sizeof(double) * 4 * std::max(n_op, 1)));
  MJRL_HIP(e, hipMalloc(&e->d_agent_body, sizeof(int32_t) * e->n_agent));
  size_t nstore = (size_t)e->n_env * e->n_agent * std::max(n_slot, 1);
  MJRL_HIP(e, hipMalloc(&e->store, sizeof(double) * nstore));
  if (n_op) {
    MJRL_HIP(e, hipMemcpy(e->d_prog_i, prog_i, sizeof(int32_t) * 8 * n_op, hipMemcpyHostToDevice));
    MJRL_HIP(e, hipMemcpy(e->d_prog_f, prog_f, sizeof(double) * 4 * n_op, hipMemcpyHostToDevice));
  }
  MJRL_HIP(e, hipMemcpy(e->d_agent_body, agent_body, sizeof(int32_t) * e->n_agent, hipMemcpyHostToDevice));
  std::vector<double> nan(nstore, __builtin_nan(""));
  MJRL_HIP(e, hipMemcpy(e->store, nan.data(), sizeof(double) * nstore, hipMemcpyHostToDevice));
  return upload_gather(e);
}

int mjrl_set_gather_tables(mjrl_env* e, int n_agent, const int32_t* n_sensor, const int32_t* sensor_idx,
                           const int32_t* n_qpos, const int32_t* qpos_idx, const int32_t* n_qvel,
                           const int32_t* qvel_idx) {
  MJRL_ENTER(e);
  if (n_agent <= 0) MJRL_FAIL(e, 1, "set_gather_tables: n_agent must be positive");
  if (e->n_agent && e->n_agent != n_agent) MJRL_FAIL(e, 1, "set_gather_tables: agent count differs from the scatter table's");
  const DevModel& m = e->hm;
  int dim = 0;
  for (int a = 0; a < n_agent; a++) dim = std::max(dim, n_sensor[a] + n_qpos[a] + n_qvel[a]);
  std::vector<int32_t> table((size_t)n_agent * dim, -1);
  int so = 0, po = 0, vo = 0;
  for (int a = 0; a < n_agent; a++) {
    int k = 0;
    for (int i = 0; i < n_sensor[a]; i++) {
      int idx = sensor_idx[so++];
      if (idx < 0 || idx >= m.nsensordata) MJRL_FAIL(e, 2, "set_gather_tables: sensordata index %d out of range", idx);
      table[(size_t)a * dim + k++] = (0 << 24) | idx;
    }
    for (int i = 0; i < n_qpos[a]; i++) {
      int idx = qpos_idx[po++];
      if (idx < 0 || idx >= m.nq) MJRL_FAIL(e, 2, "set_gather_tables: qpos index %d out of range", idx);
      table[(size_t)a * dim + k++] = (1 << 24) | idx;
    }
    for (int i = 0; i < n_qvel[a]; i++) {
      int idx = qvel_idx[vo++];
      if (idx < 0 || idx >= m.nv) MJRL_FAIL(e, 2, "set_gather_tables: qvel index %d out of range", idx);
      table[(size_t)a * dim + k++] = (2 << 24) | idx;
    }
  }
  e->h_obs_len.assign(n_agent, 0);
  for (int a = 0; a < n_agent; a++) e->h_obs_len[a] = n_sensor[a] + n_qpos[a] + n_qvel[a];
  e->h_gather = table;
  e->n_agent = n_agent;
  e->base_obs_dim = dim;
  return upload_gather(e);
}

int mjrl_set_scatter_tables(mjrl_env* e, int n_agent, int mode, const int32_t* n_idx, const int32_t* idx) {
  MJRL_ENTER(e);
  if (n_agent <= 0) MJRL_FAIL(e, 1, "set_scatter_tables: n_agent must be positive");
  if (e->n_agent && e->n_agent != n_agent) MJRL_FAIL(e, 1, "set_scatter_tables: agent count differs from the gather table's");
  const DevModel& m = e->hm;
  int limit = mode == 0 ? m.nu : m.nv;
  std::vector<std::vector<int32_t>> lists(n_agent);
  int o = 0;
  for (int a = 0; a < n_agent; a++)
    for (int i = 0; i < n_idx[a]; i++) {
      int v = idx[o++];
      if (v < 0 || v >= limit) MJRL_FAIL(e, 2, "set_scatter_tables: index %d out of range for mode %d", v, mode);
      lists[a].push_back(v);
    }
  e->h_scatter = lists;
  e->scatter_mode = mode;
  e->scatter_act_dim = -1;
  e->n_agent = n_agent;
  return 0;
}

int mjrl_set_tag_tables(mjrl_env* e, int n_tag, const int32_t* tag_num, const int32_t* tag_ref) {
  MJRL_ENTER(e);
  if (n_tag < 0) MJRL_FAIL(e, 1, "set_tag_tables: negative tag count");
  std::vector<int32_t> adr(std::max(n_tag, 1), 0), num(std::max(n_tag, 1), 0);
  int total = 0;
  for (int t = 0; t < n_tag; t++) {
    if (tag_num[t] < 0) MJRL_FAIL(e, 2, "set_tag_tables: tag %d has a negative object count", t);
    adr[t] = total; num[t] = tag_num[t]; total += tag_num[t];
  }
  for (int k = 0; k < total; k++) {
    const int kind = tag_ref[k] >> 16, id = tag_ref[k] & 0xFFFF;
    if (!((kind == 0 && id < e->hm.nbody) || (kind == 1 && id < e->hm.ngeom)) || tag_ref[k] < 0)
      MJRL_FAIL(e, 2, "set_tag_tables: entry %d (kind %d, id %d) names no body / geom of the level", k, kind, id);
  }
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  for (void** p : {(void**)&e->d_tag_adr, (void**)&e->d_tag_num, (void**)&e->d_tag_ref})
    if (*p) { hipFree(*p); *p = nullptr; }
  MJRL_HIP(e, hipMalloc(&e->d_tag_adr, sizeof(int32_t) * adr.size()));
  MJRL_HIP(e, hipMalloc(&e->d_tag_num, sizeof(int32_t) * num.size()));
  MJRL_HIP(e, hipMalloc(&e->d_tag_ref, sizeof(int32_t) * std::max(total, 1)));
  MJRL_HIP(e, hipMemcpy(e->d_tag_adr, adr.data(), sizeof(int32_t) * adr.size(), hipMemcpyHostToDevice));
  MJRL_HIP(e, hipMemcpy(e->d_tag_num, num.data(), sizeof(int32_t) * num.size(), hipMemcpyHostToDevice));
  if (total) MJRL_HIP(e, hipMemcpy(e->d_tag_ref, tag_ref, sizeof(int32_t) * total, hipMemcpyHostToDevice));
  e->n_tag = n_tag;
  e->h_tag_num.assign(num.begin(), num.begin() + n_tag);
  return 0;
}

int mjrl_set_env_base(mjrl_env* e, int first_env_id) {
  MJRL_ENTER(e);
  if (first_env_id < 0) MJRL_FAIL(e, 1, "set_env_base: negative id");
  e->env_base = first_env_id;
  return 0;
}

int mjrl_set_variants(mjrl_env* e, int n_variant, const double* rgba, unsigned long long seed) {
  MJRL_ENTER(e);
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  for (void** p : {(void**)&e->variant, (void**)&e->variant_rgba})
    if (*p) { hipFree(*p); *p = nullptr; }
  e->n_variant = 0;
  if (n_variant <= 0) return 0;
  if (!rgba) MJRL_FAIL(e, 1, "set_variants: no colour table");
  const size_t n = (size_t)n_variant * 4 * e->hm.ngeom;
  MJRL_HIP(e, hipMalloc(&e->variant, sizeof(int) * e->n_env));
  MJRL_HIP(e, hipMalloc(&e->variant_rgba, sizeof(double) * std::max<size_t>(n, 1)));
  MJRL_HIP(e, hipMemset(e->variant, 0, sizeof(int) * e->n_env));
  MJRL_HIP(e, hipMemset(e->episode, 0, sizeof(int) * e->n_env));     // (variants are drawn from episode 1 on)
  MJRL_HIP(e, hipMemcpy(e->variant_rgba, rgba, sizeof(double) * n, hipMemcpyHostToDevice));
  e->n_variant = n_variant;
  e->variant_seed = seed;
  return 0;
}

static unsigned short host_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

int mjrl_encoder_load(mjrl_env* e, int latent_dim, int relu_latent, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* wd, const float* bd) {
  MJRL_ENTER(e);
  if (latent_dim < 1 || latent_dim > 1024) MJRL_FAIL(e, 1, "encoder_load: latent_dim must be in 1..1024");
  if (!w1 || !b1 || !w2 || !b2 || !wd || !bd) MJRL_FAIL(e, 1, "encoder_load: null weights");
  if (e->n_cam_obs && latent_dim != e->enc_latent) MJRL_FAIL(e, 1, "encoder_load: camera observations are laid out for latent_dim %d", e->enc_latent);
  const int tiles = (latent_dim + 15) / 16;
  auto lane_k = [](int lane, int j) { return 8 * (lane >> 4) + j; };
  // conv1: W[n][k] scaled by 1/255 (the pixels enter as integers 0..255), K reordered so that a lane's eight k-values are
  // eight contiguous bytes of the image (mjrl_encoder.h): k = 8 ky + j is byte j = 3 kx + c of the 9-byte run of image row
  // ky, k = 24 + ky the run's ninth byte (kx = 2, c = 2); 27..31 are padding
  auto conv1_source = [](int k) {            // -> index (ky * 3 + kx) * 3 + c of the Keras kernel, -1 for padding
    if (k < 24) { const int ky = k / 8, j = k % 8; return (ky * 3 + j / 3) * 3 + j % 3; }
    if (k < 27) return ((k - 24) * 3 + 2) * 3 + 2;
    return -1;
  };
  std::vector<unsigned short> p1((size_t)2 * 64 * 8), p2((size_t)9 * 4 * 64 * 8), pd((size_t)(enc::FLAT / 32) * tiles * 64 * 8);
  for (int nt = 0; nt < 2; nt++)
    for (int lane = 0; lane < 64; lane++)
      for (int j = 0; j < 8; j++) {
        const int k = lane_k(lane, j), n = 16 * nt + (lane & 15);
        const int src = conv1_source(k);
        p1[((size_t)nt * 64 + lane) * 8 + j] = host_bf16(src >= 0 ? w1[src * 32 + n] / 255.0f : 0.0f);
      }
  for (int tap = 0; tap < 9; tap++)
    for (int nt = 0; nt < 4; nt++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 8; j++) {
          const int c = lane_k(lane, j), n = 16 * nt + (lane & 15);
          p2[(((size_t)tap * 4 + nt) * 64 + lane) * 8 + j] = host_bf16(w2[((size_t)tap * 32 + c) * 64 + n]);
        }
  for (int kk = 0; kk < enc::FLAT / 32; kk++)
    for (int nt = 0; nt < tiles; nt++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 8; j++) {
          // (element k of the activation vector as the conv kernel lays it out is Keras' flattened index a2_source(k))
          const int k = enc::a2_source(32 * kk + lane_k(lane, j)), n = 16 * nt + (lane & 15);
          pd[(((size_t)kk * tiles + nt) * 64 + lane) * 8 + j] = n < latent_dim ? host_bf16(wd[(size_t)k * latent_dim + n]) : 0;
        }
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  for (void** p : {&e->enc_w1, &e->enc_w2, &e->enc_wd, (void**)&e->enc_b1, (void**)&e->enc_b2, (void**)&e->enc_bd})
    if (*p) { hipFree(*p); *p = nullptr; }
  auto up = [&](void** dst, const void* src, size_t bytes) -> int {
    MJRL_HIP(e, hipMalloc(dst, bytes));
    MJRL_HIP(e, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
  };
  if (int rc = up(&e->enc_w1, p1.data(), p1.size() * 2)) return rc;
  if (int rc = up(&e->enc_w2, p2.data(), p2.size() * 2)) return rc;
  if (int rc = up(&e->enc_wd, pd.data(), pd.size() * 2)) return rc;
  if (int rc = up((void**)&e->enc_b1, b1, sizeof(float) * 32)) return rc;
  if (int rc = up((void**)&e->enc_b2, b2, sizeof(float) * 64)) return rc;
  if (int rc = up((void**)&e->enc_bd, bd, sizeof(float) * latent_dim)) return rc;
  e->enc_latent = latent_dim; e->enc_tiles = tiles; e->enc_relu = relu_latent ? 1 : 0;
  const int conv_lds = enc::IMG_LDS + enc::H1 * enc::H1 * enc::C1 * 2;
  MJRL_HIP(e, hipFuncSetAttribute((const void*)enc::mjrl_encoder_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, conv_lds));
  return 0;
}

// conv + dense over n_img images already in HBM; obs / obs_row: scatter of the latents into the observation rows
static int launch_encoder(mjrl_env* e, const uint8_t* d_rgb, int n_img, float* d_latent, double* d_obs, const int* d_obs_row) {
  if (!e->enc_wd) MJRL_FAIL(e, 3, "encode: no encoder weights loaded (mjrl_encoder_load)");
  if (n_img <= 0) return 0;
  const int groups = (e->enc_tiles + enc::DENSE_MAXT - 1) / enc::DENSE_MAXT, mtiles = (n_img + 31) / 32;
  if (n_img > e->enc_cap || e->enc_part_tiles != e->enc_tiles) {
    MJRL_HIP(e, hipStreamSynchronize(e->stream));
    for (void** p : {(void**)&e->enc_a2, (void**)&e->enc_part})
      if (*p) { hipFree(*p); *p = nullptr; }
    e->enc_cap = 0;
    MJRL_HIP(e, hipMalloc(&e->enc_a2, sizeof(unsigned short) * (size_t)n_img * enc::FLAT));
    // the dense layer's partial tiles [image][K split][column]
    MJRL_HIP(e, hipMalloc(&e->enc_part, sizeof(float) * (size_t)enc::DENSE_KSPLIT * n_img * e->enc_tiles * 16));
    e->enc_cap = n_img;
    e->enc_part_tiles = e->enc_tiles;
  }
  const int conv_lds = enc::IMG_LDS + enc::H1 * enc::H1 * enc::C1 * 2;
  hipLaunchKernelGGL(enc::mjrl_encoder_conv_kernel, dim3(n_img), dim3(enc::CONV_THREADS), conv_lds, e->stream, d_rgb, n_img,
                     (const enc::frag_ab*)e->enc_w1, e->enc_b1, (const enc::frag_ab*)e->enc_w2, e->enc_b2, e->enc_a2);
  MJRL_HIP(e, hipGetLastError());
  for (int g = 0; g < groups; g++) {
    const int nt0 = g * enc::DENSE_MAXT, nt = std::min((int)enc::DENSE_MAXT, e->enc_tiles - nt0);
    // (the partial buffer is laid out for THIS call's image count)
#define MJRL_DENSE(NT)                                                                                                       \
    hipLaunchKernelGGL(enc::mjrl_encoder_dense_kernel<NT>, dim3(mtiles, enc::DENSE_KSPLIT), dim3(64 * enc::DENSE_WAVES), 0,    \
                       e->stream, e->enc_a2, n_img, (const enc::frag_ab*)e->enc_wd, e->enc_tiles, nt0, e->enc_part)
    switch (nt) {
      case 1: MJRL_DENSE(1); break;
      case 2: MJRL_DENSE(2); break;
      case 3: MJRL_DENSE(3); break;
      case 4: MJRL_DENSE(4); break;
      case 5: MJRL_DENSE(5); break;
      case 6: MJRL_DENSE(6); break;
      default: MJRL_DENSE(7); break;
    }
#undef MJRL_DENSE
    MJRL_HIP(e, hipGetLastError());
  }
  hipLaunchKernelGGL(enc::mjrl_encoder_dense_finish_kernel, dim3((n_img * e->enc_latent + 255) / 256), dim3(256), 0, e->stream,
                     e->enc_part, n_img, e->enc_latent, e->enc_tiles, e->enc_bd, e->enc_relu, d_latent, d_obs, d_obs_row);
  MJRL_HIP(e, hipGetLastError());
  return 0;
}

int mjrl_encode_device(mjrl_env* e, const uint8_t* d_rgb, int n_img, float* d_latent) {
  MJRL_ENTER(e);
  if (!d_rgb || !d_latent) MJRL_FAIL(e, 3, "encode: null buffer");
  return launch_encoder(e, d_rgb, n_img, d_latent, nullptr, nullptr);
}

int mjrl_encode_host(mjrl_env* e, const uint8_t* h_rgb, int n_img, float* h_latent) {
  MJRL_ENTER(e);
  if (!h_rgb || !h_latent || n_img <= 0) MJRL_FAIL(e, 3, "encode: bad arguments");
  if (!e->enc_wd) MJRL_FAIL(e, 3, "encode: no encoder weights loaded (mjrl_encoder_load)");
  uint8_t* d_rgb = nullptr;
  float* d_lat = nullptr;
  const size_t nb = (size_t)n_img * enc::IMG * enc::IMG * 3;
  MJRL_HIP(e, hipMalloc(&d_rgb, nb));
  hipError_t he = hipMalloc(&d_lat, sizeof(float) * (size_t)n_img * e->enc_latent);
  int rc = 0;
  if (he == hipSuccess) he = hipMemcpyAsync(d_rgb, h_rgb, nb, hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess) rc = launch_encoder(e, d_rgb, n_img, d_lat, nullptr, nullptr);
  if (he == hipSuccess && !rc) he = hipMemcpyAsync(h_latent, d_lat, sizeof(float) * (size_t)n_img * e->enc_latent, hipMemcpyDeviceToHost, e->stream);
  if (he == hipSuccess && !rc) he = hipStreamSynchronize(e->stream);
  hipFree(d_rgb);
  if (d_lat) hipFree(d_lat);
  if (he != hipSuccess) { e->err = hipGetErrorString(he); return 100 + (int)he; }
  return rc;
}

static int ensure_scene(mjrl_env* e) {
  if (!e->scene) MJRL_HIP(e, hipMalloc(&e->scene, sizeof(double) * (size_t)e->n_env * scene_doubles(e->hm)));
  return 0;
}

int mjrl_set_scene_cache(mjrl_env* e, int enabled) {
  MJRL_ENTER(e);
  if (enabled && e->hm.ncam == 0) MJRL_FAIL(e, 3, "set_scene_cache: the level has no cameras");
  if (enabled) if (int rc = ensure_scene(e)) return rc;
  if ((enabled != 0) != e->scene_on) e->scene_valid = false;
  e->scene_on = enabled != 0;
  return 0;
}

int mjrl_set_camera_obs(mjrl_env* e, int n_agent, const int32_t* agent_cam) {
  MJRL_ENTER(e);
  if (n_agent == 0 || !agent_cam) {            // off
    e->n_cam_obs = 0;
    e->h_agent_cam.clear();
    return e->n_agent ? upload_gather(e) : 0;
  }
  if (!e->enc_wd) MJRL_FAIL(e, 3, "set_camera_obs: load the encoder first (mjrl_encoder_load)");
  if (!e->n_agent || e->h_obs_len.empty() || n_agent != e->n_agent) MJRL_FAIL(e, 3, "set_camera_obs: set the gather tables first (same agent count)");
  if (e->hm.ncam == 0) MJRL_FAIL(e, 3, "set_camera_obs: the level has no cameras");
  for (int a = 0; a < n_agent; a++)
    if (agent_cam[a] < -1 || agent_cam[a] >= e->hm.ncam) MJRL_FAIL(e, 2, "set_camera_obs: camera id %d out of range", agent_cam[a]);
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  if (!e->enc_rgb) MJRL_HIP(e, hipMalloc(&e->enc_rgb, (size_t)e->n_env * e->hm.ncam * enc::IMG * enc::IMG * 3));
  e->h_agent_cam.assign(agent_cam, agent_cam + n_agent);
  e->n_cam_obs = e->enc_latent;
  if (!e->scene_on) if (int rc = mjrl_set_scene_cache(e, 1)) return rc;      // (the step hands its frames to the ray caster)
  return upload_gather(e);
}

int mjrl_size(const mjrl_env* e, const char* name) {
#define X(field) if (strcmp(name, #field) == 0) return e->hm.field;
  MJRL_SIZE_FIELDS(X)
#undef X
  if (!strcmp(name, "obs_dim")) return e->obs_dim;
  if (!strcmp(name, "n_agent")) return e->n_agent;
  if (!strcmp(name, "n_env")) return e->n_env;
  // which specialised build fits the batch (kernel_cache.code_object): the one for few copies only where it differs --
  // a roomy model's kernel holds the big solver forms at any batch size
  if (!strcmp(name, "few")) return (e->few && !mj::pgs_roomy(e->hm, e->lay)) ? 1 : 0;
  if (!strcmp(name, "blocks_per_cu")) {      // what the runtime says about residency of the step kernel
    int n = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, e->big ? (const void*)mjrl_step_kernel_big : (const void*)mjrl_step_kernel, 64,
                                                     (size_t)e->lay.total * sizeof(double)) != hipSuccess) return -1;
    return n;
  }
  if (!strcmp(name, "n_slot")) return e->n_slot;
  if (!strcmp(name, "n_tag")) return e->n_tag;
  if (!strcmp(name, "latent_dim")) return e->enc_latent;
  if (!strcmp(name, "n_camera_obs")) return e->n_cam_obs;
  if (!strcmp(name, "n_variant")) return e->n_variant;
  if (!strcmp(name, "env_base")) return e->env_base;
  if (!strcmp(name, "n_extra_obs")) return e->n_extra;
  if (!strcmp(name, "lds_doubles")) return e->lay.total;
  if (!strcmp(name, "con_stride")) return mj::CON_STRIDE;
  if (!strcmp(name, "row_stride")) return mj::ROW_STRIDE;
  if (!strcmp(name, "ldj")) return e->lay.ldj;
  return -1;
}

int mjrl_lds_offset(const mjrl_env* e, const char* region) {
  const mj::Lay& l = e->lay;
#define R(name) if (!strcmp(region, #name)) return l.name;
  R(qpos) R(qvel) R(ctrl) R(warm) R(xpos) R(xquat) R(xanchor) R(xaxis) R(com) R(cinert) R(crb) R(cdof) R(cdofdot)
  R(cvel) R(cacc) R(LD) R(Dinv) R(gpos) R(gquat) R(bias) R(smooth) R(qaccs) R(x) R(qfc) R(qacc) R(con) R(J) R(row)
  R(sens) R(ints) R(total) R(i_item) R(i_cong1) R(i_cong2) R(i_conadr) R(i_rowid) R(i_rowinfo)
#undef R
  return -1;
}

static int ensure_scatter(mjrl_env* e, int act_dim) {
  if (e->h_scatter.empty() || e->scatter_act_dim == act_dim) return 0;
  std::vector<int32_t> table((size_t)e->n_agent * act_dim, -1);
  for (int a = 0; a < e->n_agent; a++) {
    if ((int)e->h_scatter[a].size() > act_dim)
      MJRL_FAIL(e, 3, "step: agent %d routes %d action slots to the physics but act_dim is %d", a,
                (int)e->h_scatter[a].size(), act_dim);
    for (size_t k = 0; k < e->h_scatter[a].size(); k++) table[(size_t)a * act_dim + k] = e->h_scatter[a][k];
  }
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  if (e->d_scatter) { hipFree(e->d_scatter); e->d_scatter = nullptr; }
  MJRL_HIP(e, hipMalloc(&e->d_scatter, sizeof(int32_t) * std::max<size_t>(table.size(), 1)));
  MJRL_HIP(e, hipMemcpy(e->d_scatter, table.data(), sizeof(int32_t) * table.size(), hipMemcpyHostToDevice));
  e->scatter_act_dim = act_dim;
  return 0;
}

// mode: 0 = full step; 1 = forward pass only (no integration, no counters): reset observations and queries
static int launch_step(mjrl_env* e, const double* d_actions, int act_dim, int skip_frames, double* d_obs, double* d_reward,
                       uint8_t* d_term, uint8_t* d_trunc, double* d_dbg, int dbg_stage, int forward_only,
                       unsigned long long* d_stamps, unsigned long long* d_timeline) {
  if (skip_frames < 0) MJRL_FAIL(e, 3, "step: skip_frames must be >= 0");
  if (d_obs && !e->d_gather) MJRL_FAIL(e, 3, "step: observations requested but no gather table is set");
  mj::StepArgs a{};
  a.qpos = e->qpos; a.qvel = e->qvel; a.ctrl = e->ctrl; a.warm = e->warm; a.sensordata = e->sens;
  a.overflow = e->overflow;
  a.stats = e->stats;
  a.timestep = e->timestep;
  a.actions = nullptr; a.scatter = nullptr;
  a.n_agent = e->n_agent; a.act_dim = act_dim; a.scatter_mode = e->scatter_mode;
  if (d_actions && !e->h_scatter.empty()) {
    if (int rc = ensure_scatter(e, act_dim)) return rc;
    a.actions = d_actions;
    a.scatter = e->d_scatter;
  }
  a.gather = e->d_gather; a.obs_dim = e->obs_dim; a.obs = d_obs;
  a.reward = d_reward; a.term = d_term; a.trunc = d_trunc;
  a.max_steps = e->max_steps; a.n_env = e->n_env;
  a.dbg_stage = dbg_stage;
  a.forward_only = forward_only;
  a.stamps = d_stamps;
  a.timeline = d_timeline;
  a.tag_adr = e->d_tag_adr; a.tag_num = e->d_tag_num; a.tag_ref = e->d_tag_ref;
  a.env_base = e->env_base;
  a.variant = e->variant; a.episode = e->episode; a.n_variant = e->n_variant; a.variant_seed = e->variant_seed;
  a.reset_mask = forward_only ? nullptr : e->step_reset_mask;
  a.reset_warm = e->reset_warm;
  a.reset_sens = e->reset_sens;
  a.reset_scene = e->reset_scene;
  a.few = e->few ? 1 : 0;
  a.lane_rec = e->d_lane_rec;
  a.io_agent1 = e->io_agent + 1; a.obs_f32 = e->obs_f32;
  if (e->io_agent >= 0 || e->obs_f32) {
    if (e->n_cam_obs) MJRL_FAIL(e, 3, "step: camera latents in the observation need the default I/O layout (mjrl_set_io_layout)");
    if (e->io_agent >= e->n_agent) MJRL_FAIL(e, 3, "step: I/O layout names agent %d of %d", e->io_agent, e->n_agent);
    if (e->io_agent >= 0) {
      // (the one-agent action row is read lane by lane; a fused program must be one the kernel stages in LDS)
      const int n_act_row = e->n_agent * act_dim, n_store_row = e->n_agent * e->n_slot, n_op = forward_only ? 0 : e->n_op;
      const bool staged = n_op == 0 || (n_act_row + n_store_row + 8 * n_op + e->n_agent <= 4 * e->hm.nv && 8 * n_op <= 64);
      if (n_act_row > 64 || !staged)
        MJRL_FAIL(e, 3, "step: the one-agent I/O layout needs n_agent x act_dim <= 64 and a fused program small enough to be staged");
    }
  }
  a.auto_mask = forward_only ? nullptr : e->auto_mask;
  a.auto_mode = e->auto_mode;
  a.prog_i = e->d_prog_i; a.prog_f = e->d_prog_f; a.n_op = forward_only ? 0 : e->n_op; a.n_slot = e->n_slot;
  a.agent_body = e->d_agent_body; a.agent_obs_len = e->d_obs_len; a.store = e->store;
  if (e->n_op && !forward_only && !a.actions && d_actions) a.actions = d_actions;   // ops read their action slots
  if (e->frames && skip_frames > 0) e->frames_valid = true;
  if (e->scene_on && skip_frames > 0) e->scene_valid = true;
  size_t lds_bytes = (size_t)e->lay.total * sizeof(double);
  if (const char* pad = getenv("MJRL_LDS_BYTES")) lds_bytes = std::max(lds_bytes, (size_t)atoi(pad));   // experiments: residency
  // The kernel advances one physics frame; a step of skipFrames frames (mujoco_parent.py:333-336) is that many launches
  // on the stream.  The action scatter belongs to the first, everything after the physics (frame cache, debug dump,
  // counters, observations, plugin ops) to the last.
  // (a Runge-Kutta frame is four launches, one forward pass each)
  const int passes = (e->hm.integrator == 1 && !forward_only) ? 4 : 1;
  const int launches = skip_frames > 0 ? skip_frames * passes : 1;
  a.rk = e->rk;
  for (int f = 0; f < launches; f++) {
    const bool last = f == launches - 1;
    a.rk_stage = f % passes;
    a.skip_frames = skip_frames > 0 ? 1 : 0;
    a.more_frames = last ? 0 : 1;
    if (f > 0) a.scatter = nullptr;
    a.first_frame = f == 0;
    a.dbg = last ? d_dbg : nullptr;
    a.frames = last ? e->frames : nullptr;
    a.scene = (last && e->scene_on) ? e->scene : nullptr;
    a.lpt_count_in = nullptr; a.lpt_mask_in = nullptr; a.lpt_count_out = nullptr; a.lpt_mask_out = nullptr;
    a.lpt_count_clear = nullptr; a.lpt_mask_clear = nullptr;
    a.lpt_words = e->lpt_words;
    a.stop_after = e->stop_after;
    if (e->lpt_enabled && !forward_only && !d_dbg && !e->stop_after) {
      int in = e->lpt_cur, out = (e->lpt_cur + 1) % 3, next = (e->lpt_cur + 2) % 3;
      a.lpt_count_clear = e->lpt_count[next];
      a.lpt_mask_clear = e->lpt_mask[next];
      if (e->lpt_valid) { a.lpt_count_in = e->lpt_count[in]; a.lpt_mask_in = e->lpt_mask[in]; }
      a.lpt_count_out = e->lpt_count[out];
      a.lpt_mask_out = e->lpt_mask[out];
      e->lpt_cur = out;
      e->lpt_valid = true;
    }
    // launches that use a diagnostic (LDS dump, stage clock, timeline, stage cut) need a build that has them
    const bool diag = a.dbg || a.stamps || a.timeline || a.stop_after;
    if (e->spec_fn && (!diag || e->spec_diag)) {
      const void* image = e->d_blob;        // the specialised kernel derives every section from the image base
      void* params[] = {(void*)&image, (void*)&a};
      MJRL_HIP(e, hipModuleLaunchKernel(e->spec_fn, e->n_env, 1, 1, 64, 1, 1, (unsigned)lds_bytes, e->stream, params, nullptr));
    } else if (diag) {
      if (e->big) hipLaunchKernelGGL(mjrl_step_kernel_big_diag, dim3(e->n_env), dim3(64), lds_bytes, e->stream, e->d_model, a);
      else hipLaunchKernelGGL(mjrl_step_kernel_diag, dim3(e->n_env), dim3(64), lds_bytes, e->stream, e->d_model, a);
      MJRL_HIP(e, hipGetLastError());
    } else {
      if (e->big) hipLaunchKernelGGL(mjrl_step_kernel_big, dim3(e->n_env), dim3(64), lds_bytes, e->stream, e->d_model, a);
      else hipLaunchKernelGGL(mjrl_step_kernel, dim3(e->n_env), dim3(64), lds_bytes, e->stream, e->d_model, a);
      MJRL_HIP(e, hipGetLastError());
    }
  }
  // Camera observations (mjrl_set_camera_obs): the agents' cameras are rendered at the new state and encoded, the
  // latents land in the observation rows' last slots -- same stream, behind the step.
  if (e->n_cam_obs && d_obs && !forward_only) {
    if (int rc = mjrl_render_device(e, enc::IMG, enc::IMG, e->enc_rgb)) return rc;
    if (int rc = launch_encoder(e, e->enc_rgb, e->n_env * e->hm.ncam, nullptr, d_obs, e->enc_obs_row)) return rc;
  }
  return 0;
}

int mjrl_set_io_layout(mjrl_env* e, int agent, int obs_f32) {
  MJRL_ENTER(e);
  if (agent < -1 || (e->n_agent && agent >= e->n_agent)) MJRL_FAIL(e, 2, "set_io_layout: agent %d out of range", agent);
  if ((agent >= 0 || obs_f32) && e->n_cam_obs)
    MJRL_FAIL(e, 3, "set_io_layout: camera latents in the observation need the default layout");
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  e->io_agent = agent;
  e->obs_f32 = obs_f32 ? 1 : 0;
  return 0;
}

int mjrl_set_autoreset(mjrl_env* e, int mode) {
  MJRL_ENTER(e);
  if (mode < 0 || mode > 2) MJRL_FAIL(e, 4, "set_autoreset: mode %d (0 off, 1 reset without a step, 2 reset then step)", mode);
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  if (mode == 0) {
    if (e->auto_mask) { hipFree(e->auto_mask); e->auto_mask = nullptr; }
    e->auto_mode = 0;
    return 0;
  }
  if (!e->auto_mask) MJRL_HIP(e, hipMalloc(&e->auto_mask, (size_t)e->n_env));
  MJRL_HIP(e, hipMemset(e->auto_mask, 0, (size_t)e->n_env));
  e->auto_mode = mode;
  return 0;
}

int mjrl_cap_overflows(mjrl_env* e, unsigned long long* h_counts, int clear) {
  MJRL_ENTER(e);
  if (!h_counts) MJRL_FAIL(e, 4, "cap_overflows: null output");
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  unsigned long long all[3] = {0, 0, 0};
  MJRL_HIP(e, hipMemcpy(all, e->overflow, sizeof(all), hipMemcpyDeviceToHost));
  h_counts[0] = all[0]; h_counts[1] = all[1];
  if (clear) MJRL_HIP(e, hipMemset(e->overflow, 0, sizeof(unsigned long long) * 2));
  // (never seen: a workgroup whose share of the dispatch tables held no copy left its copy unstepped)
  if (all[2]) MJRL_FAIL(e, 7, "dispatch tables were inconsistent in %llu workgroup launches: copies were not stepped", all[2]);
  return 0;
}

int mjrl_load_kernel(mjrl_env* e, const char* path) {
  MJRL_ENTER(e);
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  if (e->spec_module) { hipModuleUnload(e->spec_module); e->spec_module = nullptr; e->spec_fn = nullptr; e->spec_diag = false; }
  if (!path) return 0;                      // back to the generic kernel
  hipModule_t mod = nullptr;
  if (hipModuleLoad(&mod, path) != hipSuccess) MJRL_FAIL(e, 6, "load_kernel: cannot load code object %s", path);
  hipFunction_t fn = nullptr;
  hipDeviceptr_t d_sizes = nullptr;
  size_t nb = 0;
  if (hipModuleGetFunction(&fn, mod, "mjrl_step_kernel_spec") != hipSuccess ||
      hipModuleGetGlobal(&d_sizes, &nb, mod, "mjrl_spec_sizes") != hipSuccess || nb != sizeof(int) * MJRL_NSIZES) {
    hipModuleUnload(mod);
    MJRL_FAIL(e, 6, "load_kernel: %s does not export mjrl_step_kernel_spec / mjrl_spec_sizes", path);
  }
  int built[MJRL_NSIZES];
  const int32_t* want = (const int32_t*)(e->h_blob.data() + 8);
  if (hipMemcpy(built, d_sizes, nb, hipMemcpyDeviceToHost) != hipSuccess || memcmp(built, want, nb) != 0) {
    hipModuleUnload(mod);
    MJRL_FAIL(e, 6, "load_kernel: %s was built for a different model shape", path);
  }
  // A code object is built apart from the library: it must agree with it on the layout of StepArgs, on where the kernels
  // read it in the kernarg segment and on the kernel sources (mjrl_step.h, MJRL_SPEC_ABI_INIT).  An object that disagrees
  // would read garbage pointers or step another revision's physics -- and be timed without complaint.
  {
    hipDeviceptr_t d_abi = nullptr;
    size_t na = 0;
    unsigned long long abi[mj::SPEC_ABI_WORDS] = {0, 0, 0};
    if (hipModuleGetGlobal(&d_abi, &na, mod, "mjrl_spec_abi") != hipSuccess || na != sizeof(abi) ||
        hipMemcpy(abi, d_abi, na, hipMemcpyDeviceToHost) != hipSuccess) {
      hipModuleUnload(mod);
      (void)hipGetLastError();
      MJRL_FAIL(e, 6, "load_kernel: %s carries no mjrl_spec_abi record (built before the library's revision)", path);
    }
    if (abi[0] != LIB_ABI[0] || abi[2] != LIB_ABI[2]) {
      hipModuleUnload(mod);
      MJRL_FAIL(e, 6, "load_kernel: %s was built with another layout of the step arguments (StepArgs)", path);
    }
    // (experiments that A/B a saved object of an older revision against this library say so: MJRL_ALLOW_STALE_KERNEL=1)
    const char* stale_ok = getenv("MJRL_ALLOW_STALE_KERNEL");
    if (abi[1] != LIB_ABI[1] && !(stale_ok && atoi(stale_ok))) {
      hipModuleUnload(mod);
      MJRL_FAIL(e, 6, "load_kernel: %s was built from other kernel sources than this library (digest %012llx, library "
                      "%012llx); rebuild it (kernel_cache.code_object)", path, abi[1], LIB_ABI[1]);
    }
    hipFunction_t check = nullptr;
    int* d_ok = nullptr;
    int ok = 0;
    bool ran = hipModuleGetFunction(&check, mod, "mjrl_spec_selfcheck") == hipSuccess &&
               hipMalloc(&d_ok, sizeof(int)) == hipSuccess && hipMemset(d_ok, 0, sizeof(int)) == hipSuccess;
    if (ran) {
      const void* image = e->d_blob;
      mj::StepArgs probe = selfcheck_args();
      void* params[] = {(void*)&image, (void*)&probe, (void*)&d_ok};
      ran = hipModuleLaunchKernel(check, 1, 1, 1, 64, 1, 1, 0, e->stream, params, nullptr) == hipSuccess &&
            hipMemcpyAsync(&ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, e->stream) == hipSuccess &&
            hipStreamSynchronize(e->stream) == hipSuccess;
    }
    if (d_ok) hipFree(d_ok);
    if (!ran || !ok) {
      hipModuleUnload(mod);
      (void)hipGetLastError();
      MJRL_FAIL(e, 6, "load_kernel: %s failed the kernarg offset self-check (its kernel would read its arguments in the "
                      "wrong place)", path);
    }
  }
  size_t lds_bytes = (size_t)e->lay.total * sizeof(double);
  if (lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
    hipModuleUnload(mod);
    MJRL_FAIL(e, 5, "load_kernel: cannot raise the dynamic LDS limit to %zu bytes", lds_bytes);
  }
  e->spec_module = mod;
  e->spec_fn = fn;
  // (a code object built with -DMJRL_DIAG says so; without the symbol it is a production build)
  hipDeviceptr_t d_diag = nullptr;
  size_t nd = 0;
  int has_diag = 0;
  if (hipModuleGetGlobal(&d_diag, &nd, mod, "mjrl_spec_diag") == hipSuccess && nd == sizeof(int))
    if (hipMemcpy(&has_diag, d_diag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) has_diag = 0;
  e->spec_diag = has_diag != 0;
  (void)hipGetLastError();      // (a production build has no such symbol: the failed lookup must not linger as the last error)
  return 0;
}

int mjrl_step_device(mjrl_env* e, const double* d_actions, int act_dim, int skip_frames, double* d_obs, double* d_reward,
                     uint8_t* d_term, uint8_t* d_trunc) {
  MJRL_ENTER(e);
  return launch_step(e, d_actions, act_dim, skip_frames, d_obs, d_reward, d_term, d_trunc, nullptr, 0, 0);
}

static int ensure_staging(mjrl_env* e, int act_dim) {
  size_t act_n = (size_t)e->n_env * std::max(e->n_agent, 1) * std::max(act_dim, 1);
  size_t obs_n = (size_t)e->n_env * std::max(e->n_agent, 1) * std::max(e->obs_dim, 1);
  if (act_n > e->s_act_n) {
    if (e->s_act) hipFree(e->s_act);
    MJRL_HIP(e, hipMalloc(&e->s_act, sizeof(double) * act_n));
    e->s_act_n = act_n;
  }
  if (obs_n > e->s_obs_n) {
    if (e->s_obs) hipFree(e->s_obs);
    MJRL_HIP(e, hipMalloc(&e->s_obs, sizeof(double) * obs_n));
    e->s_obs_n = obs_n;
  }
  size_t na = (size_t)e->n_env * std::max(e->n_agent, 1);
  if (!e->s_rew) {
    MJRL_HIP(e, hipMalloc(&e->s_rew, sizeof(double) * na));
    MJRL_HIP(e, hipMalloc(&e->s_term, na));
    MJRL_HIP(e, hipMalloc(&e->s_trunc, na));
  }
  return 0;
}

int mjrl_step_host(mjrl_env* e, const double* h_actions, int act_dim, int skip_frames, double* h_obs, double* h_reward,
                   uint8_t* h_term, uint8_t* h_trunc) {
  MJRL_ENTER(e);
  if (e->io_agent >= 0 || e->obs_f32)
    MJRL_FAIL(e, 3, "step_host copies buffers of the default layout; with mjrl_set_io_layout use mjrl_step_pinned / mjrl_step_device");
  if (int rc = ensure_staging(e, act_dim)) return rc;
  size_t na = (size_t)e->n_env * std::max(e->n_agent, 1);
  if (h_actions)
    MJRL_HIP(e, hipMemcpyAsync(e->s_act, h_actions, sizeof(double) * na * act_dim, hipMemcpyHostToDevice, e->stream));
  int rc = launch_step(e, h_actions ? e->s_act : nullptr, act_dim, skip_frames, (h_obs && e->d_gather) ? e->s_obs : nullptr,
                       e->s_rew, e->s_term, e->s_trunc, nullptr, 0, 0);
  if (rc) return rc;
  if (h_obs && e->d_gather)
    MJRL_HIP(e, hipMemcpyAsync(h_obs, e->s_obs, sizeof(double) * na * e->obs_dim, hipMemcpyDeviceToHost, e->stream));
  if (h_reward) MJRL_HIP(e, hipMemcpyAsync(h_reward, e->s_rew, sizeof(double) * na, hipMemcpyDeviceToHost, e->stream));
  if (h_term) MJRL_HIP(e, hipMemcpyAsync(h_term, e->s_term, na, hipMemcpyDeviceToHost, e->stream));
  if (h_trunc) MJRL_HIP(e, hipMemcpyAsync(h_trunc, e->s_trunc, na, hipMemcpyDeviceToHost, e->stream));
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  return 0;
}

// Pinned host buffers the step kernel reads and writes in place: the observation rows travel over PCIe as the waves
// that produce them finish, under the rest of the launch, instead of in copies after it.
int mjrl_host_buffers(mjrl_env* e, int act_dim, double** h_actions, double** h_obs, double** h_reward, uint8_t** h_term,
                      uint8_t** h_trunc) {
  MJRL_ENTER(e);
  if (act_dim < 0) MJRL_FAIL(e, 4, "host_buffers: act_dim %d", act_dim);
  const size_t na = (size_t)e->n_env * std::max(e->n_agent, 1);
  const size_t act_n = na * std::max(act_dim, 1), obs_n = na * std::max(e->obs_dim, 1);
  if (act_n > e->p_act_n) {
    if (e->p_act) hipHostFree(e->p_act);
    e->p_act = nullptr; e->p_act_n = 0;
    MJRL_HIP(e, hipHostMalloc((void**)&e->p_act, sizeof(double) * act_n, hipHostMallocMapped));
    memset(e->p_act, 0, sizeof(double) * act_n);
    e->p_act_n = act_n;
  }
  if (obs_n > e->p_obs_n) {
    if (e->p_obs) hipHostFree(e->p_obs);
    e->p_obs = nullptr; e->p_obs_n = 0;
    MJRL_HIP(e, hipHostMalloc((void**)&e->p_obs, sizeof(double) * obs_n, hipHostMallocMapped));
    memset(e->p_obs, 0, sizeof(double) * obs_n);
    e->p_obs_n = obs_n;
  }
  // rewards and flags: one element per (copy, agent).  Re-sized when the agent count grew since the last call (a first
  // call before the gather / scatter tables exist sees n_agent = 0), and handed out only when all three exist: the
  // kernel writes n_env x n_agent elements through these pointers into host memory.
  if (na > e->p_na) {
    void* old3[] = {e->p_rew, e->p_term, e->p_trunc};
    for (void* q : old3) if (q) hipHostFree(q);
    e->p_rew = nullptr; e->p_term = nullptr; e->p_trunc = nullptr; e->p_na = 0;
    double* rew = nullptr;
    unsigned char *term = nullptr, *trunc = nullptr;
    hipError_t he = hipHostMalloc((void**)&rew, sizeof(double) * na, hipHostMallocMapped);
    if (he == hipSuccess) he = hipHostMalloc((void**)&term, na, hipHostMallocMapped);
    if (he == hipSuccess) he = hipHostMalloc((void**)&trunc, na, hipHostMallocMapped);
    if (he != hipSuccess) {
      if (rew) hipHostFree(rew);
      if (term) hipHostFree(term);
      if (trunc) hipHostFree(trunc);
      MJRL_FAIL(e, 100 + (int)he, "host_buffers: hipHostMalloc of the reward / flag buffers failed: %s", hipGetErrorString(he));
    }
    memset(rew, 0, sizeof(double) * na); memset(term, 0, na); memset(trunc, 0, na);
    e->p_rew = rew; e->p_term = term; e->p_trunc = trunc; e->p_na = na;
  }
  if (h_actions) *h_actions = e->p_act;
  if (h_obs) *h_obs = e->p_obs;
  if (h_reward) *h_reward = e->p_rew;
  if (h_term) *h_term = e->p_term;
  if (h_trunc) *h_trunc = e->p_trunc;
  return 0;
}

int mjrl_step_pinned(mjrl_env* e, int act_dim, int skip_frames) {
  MJRL_ENTER(e);
  const size_t na = (size_t)e->n_env * std::max(e->n_agent, 1);
  if (!e->p_rew || !e->p_term || !e->p_trunc || na > e->p_na || na * std::max(act_dim, 1) > e->p_act_n ||
      na * std::max(e->obs_dim, 1) > e->p_obs_n)
    MJRL_FAIL(e, 4, "step_pinned: call mjrl_host_buffers with this act_dim (and after the gather tables are set) first");
  // (the buffers' device addresses: looked up when a buffer is new, not five runtime calls per step)
  if (e->p_dev_of[0] != e->p_act || e->p_dev_of[1] != e->p_obs || e->p_dev_of[2] != e->p_rew || e->p_dev_of[3] != e->p_term ||
      e->p_dev_of[4] != e->p_trunc) {
    void* host[5] = {e->p_act, e->p_obs, e->p_rew, e->p_term, e->p_trunc};
    for (int k = 0; k < 5; k++) {
      MJRL_HIP(e, hipHostGetDevicePointer(&e->p_dev[k], host[k], 0));
      e->p_dev_of[k] = host[k];
    }
  }
  void *d_act = e->p_dev[0], *d_obs = e->p_dev[1], *d_rew = e->p_dev[2], *d_term = e->p_dev[3], *d_trunc = e->p_dev[4];
  int rc = launch_step(e, act_dim > 0 ? (const double*)d_act : nullptr, act_dim, skip_frames, e->d_gather ? (double*)d_obs : nullptr,
                       (double*)d_rew, (uint8_t*)d_term, (uint8_t*)d_trunc, nullptr, 0, 0);
  if (rc) return rc;
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  return 0;
}

int mjrl_reset(mjrl_env* e, const uint8_t* h_mask, double* d_obs) {
  MJRL_ENTER(e);
  const unsigned char* d_mask = nullptr;
  if (h_mask) {
    MJRL_HIP(e, hipMemcpyAsync(e->d_mask, h_mask, e->n_env, hipMemcpyHostToDevice, e->stream));
    d_mask = e->d_mask;
  }
  e->frames_valid = false;
  if (!d_mask && e->scene_on) e->scene_valid = true;        // (every row is the reset image's)
  // mj_resetData + mj_forward (mujoco_parent.py:349-350) from the reset image; unselected copies are not touched
  return launch_reset(e, d_mask, d_obs);
}

int mjrl_reset_device(mjrl_env* e, const uint8_t* d_mask, double* d_obs) {
  MJRL_ENTER(e);
  e->frames_valid = false;
  if (!d_mask && e->scene_on) e->scene_valid = true;
  return launch_reset(e, d_mask, d_obs);
}

int mjrl_set_step_reset_mask(mjrl_env* e, const uint8_t* d_mask) {
  MJRL_ENTER(e);
  e->step_reset_mask = d_mask;
  return 0;
}

struct field_ref { void* ptr; size_t bytes; };
static int find_field(mjrl_env* e, const char* name, field_ref* f) {
  const DevModel& m = e->hm;
  size_t n = e->n_env;
  if (!strcmp(name, "qpos")) *f = {e->qpos, sizeof(double) * n * m.nq};
  else if (!strcmp(name, "qvel")) *f = {e->qvel, sizeof(double) * n * m.nv};
  else if (!strcmp(name, "ctrl")) *f = {e->ctrl, sizeof(double) * n * m.nu};
  else if (!strcmp(name, "qacc_warmstart")) *f = {e->warm, sizeof(double) * n * m.nv};
  else if (!strcmp(name, "sensordata")) *f = {e->sens, sizeof(double) * n * m.nsensordata};
  else if (!strcmp(name, "timestep")) *f = {e->timestep, sizeof(int) * n};
  else if (!strcmp(name, "solver_stats")) *f = {e->stats, sizeof(int) * 4 * n};
  else if (!strcmp(name, "variant") && e->variant) *f = {e->variant, sizeof(int) * n};
  else if (!strcmp(name, "episode") && e->episode) *f = {e->episode, sizeof(int) * n};
  else if (!strcmp(name, "store")) *f = {e->store, sizeof(double) * n * e->n_agent * e->n_slot};
  else MJRL_FAIL(e, 4, "unknown field '%s'", name);
  return 0;
}

int mjrl_get_field(mjrl_env* e, const char* name, void* h_out, size_t nbytes) {
  MJRL_ENTER(e);
  field_ref f;
  if (int rc = find_field(e, name, &f)) return rc;
  if (nbytes != f.bytes) MJRL_FAIL(e, 4, "get_field(%s): buffer holds %zu bytes, field has %zu", name, nbytes, f.bytes);
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  if (f.bytes) MJRL_HIP(e, hipMemcpy(h_out, f.ptr, f.bytes, hipMemcpyDeviceToHost));
  return 0;
}

int mjrl_render_device(mjrl_env* e, int width, int height, uint8_t* d_rgb) {
  MJRL_ENTER(e);
  if (width <= 0 || height <= 0 || !d_rgb) MJRL_FAIL(e, 3, "render: bad arguments");
  if (e->hm.ncam == 0) MJRL_FAIL(e, 3, "render: the level has no cameras");
  if (int rc = ensure_scene(e)) return rc;
  // With the scene cache on, the rows are the ones the copies' last forward pass left (a step's: one integration older
  // than qpos, what mjv_updateScene finds in MjData, mujoco_parent.py:533).  Without it, or after a state was written by
  // hand, the frames are computed from the current qpos.
  if (!(e->scene_on && e->scene_valid)) {
    hipLaunchKernelGGL(mjrl_camera_frames_kernel, dim3(e->n_env), dim3(64), (size_t)e->lay.total * sizeof(double), e->stream, e->dm,
                       e->qpos, e->n_env, e->scene);
    MJRL_HIP(e, hipGetLastError());
  }
  // groups of 16 x 8 pixel blocks per camera: enough workgroups to fill the chip's wave slots a few times over
  const int nblock = ((width + 15) / 16) * ((height + 7) / 8);
  // (round 3, 8 x 8 blocks: 2 / 4 / 8 / 16 x 2048 measured 0.247 / 0.228 / 0.218 / 0.224 ms per config-5 step; round 4,
  // 16 x 8 blocks of two rays per lane: 4 x 2048 -- four blocks per wave at 512 copies -- 93 us per launch, 8 x 2048 97)
  int target = 4 * 2048;
  if (const char* tv = getenv("MJRL_RENDER_TARGET")) target = atoi(tv) * 2048;       // (experiments)
  int tiles = (int)((target + (size_t)e->n_env * e->hm.ncam - 1) / ((size_t)e->n_env * e->hm.ncam));
  tiles = std::max(1, std::min(tiles, std::max(1, nblock / 2)));
  // (tests: MJRL_RENDER_LOOSE=1 leaves the boxes to the bounding-sphere culls alone -- the images must not change)
  const char* lv = getenv("MJRL_RENDER_LOOSE");
  const int loose = lv && atoi(lv) ? 1 : 0;
  hipLaunchKernelGGL(mjrl_render_kernel, dim3(e->n_env, e->hm.ncam * tiles), dim3(64), render_lds_bytes(e->hm), e->stream, e->dm,
                     e->scene, e->n_env, width, height, tiles, d_rgb, e->variant, e->variant_rgba, e->render_consts, loose);
  MJRL_HIP(e, hipGetLastError());
  return 0;
}

int mjrl_render_host(mjrl_env* e, int width, int height, uint8_t* h_rgb) {
  MJRL_ENTER(e);
  size_t n = (size_t)e->n_env * e->hm.ncam * width * height * 3;
  uint8_t* d = nullptr;
  MJRL_HIP(e, hipMalloc(&d, n ? n : 1));
  int rc = mjrl_render_device(e, width, height, d);
  if (!rc) {
    hipError_t he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipMemcpy(h_rgb, d, n, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { e->err = hipGetErrorString(he); rc = 100 + (int)he; }
  }
  hipFree(d);
  return rc;
}

int mjrl_set_query_cache(mjrl_env* e, int enabled) {
  MJRL_ENTER(e);
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  if (enabled && !e->frames)
    MJRL_HIP(e, hipMalloc(&e->frames, sizeof(double) * (size_t)e->n_env * mj::frame_doubles(e->hm)));
  if (!enabled && e->frames) { hipFree(e->frames); e->frames = nullptr; }
  e->frames_valid = false;
  return 0;
}

int mjrl_set_field(mjrl_env* e, const char* name, const void* h_in, size_t nbytes) {
  MJRL_ENTER(e);
  field_ref f;
  if (int rc = find_field(e, name, &f)) return rc;
  e->frames_valid = false;      // the state the cached frames belong to is being replaced
  e->scene_valid = false;
  if (nbytes != f.bytes) MJRL_FAIL(e, 4, "set_field(%s): buffer holds %zu bytes, field has %zu", name, nbytes, f.bytes);
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  if (f.bytes) MJRL_HIP(e, hipMemcpy(f.ptr, h_in, f.bytes, hipMemcpyHostToDevice));
  return 0;
}

static int ensure_dbg(mjrl_env* e) {
  if (!e->dbg) MJRL_HIP(e, hipMalloc(&e->dbg, sizeof(double) * (size_t)e->n_env * e->lay.total));
  return 0;
}

int mjrl_step_debug(mjrl_env* e, const double* d_actions, int act_dim, int skip_frames, int stage, double* h_dump,
                    size_t nbytes) {
  MJRL_ENTER(e);
  size_t need = sizeof(double) * (size_t)e->n_env * e->lay.total;
  if (nbytes != need) MJRL_FAIL(e, 4, "step_debug: dump buffer holds %zu bytes, need %zu", nbytes, need);
  if (int rc = ensure_dbg(e)) return rc;
  if (int rc = launch_step(e, d_actions, act_dim, skip_frames, nullptr, nullptr, nullptr, nullptr, e->dbg, stage, 0)) return rc;
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  MJRL_HIP(e, hipMemcpy(h_dump, e->dbg, need, hipMemcpyDeviceToHost));
  return 0;
}

int mjrl_step_profile(mjrl_env* e, const double* d_actions, int act_dim, int skip_frames, unsigned long long* h_cycles,
                      int n) {
  MJRL_ENTER(e);
  if (n != mj::N_STAMPS) MJRL_FAIL(e, 4, "step_profile: expected room for %d stage counters, got %d", (int)mj::N_STAMPS, n);
  unsigned long long* d = nullptr;
  MJRL_HIP(e, hipMalloc(&d, sizeof(unsigned long long) * n));
  MJRL_HIP(e, hipMemsetAsync(d, 0, sizeof(unsigned long long) * n, e->stream));
  int rc = launch_step(e, d_actions, act_dim, skip_frames, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, d);
  if (!rc) {
    hipError_t he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipMemcpy(h_cycles, d, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { e->err = hipGetErrorString(he); rc = 100 + (int)he; }
  }
  hipFree(d);
  return rc;
}

int mjrl_step_truncated(mjrl_env* e, int stop_after) {
  MJRL_ENTER(e);
  if (stop_after < 1 || stop_after > mj::N_STAMPS) MJRL_FAIL(e, 4, "step_truncated: stage index %d outside 1..%d", stop_after, (int)mj::N_STAMPS);
  e->stop_after = stop_after;
  int rc = launch_step(e, nullptr, 0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0);
  e->stop_after = 0;
  return rc;
}

int mjrl_step_timeline(mjrl_env* e, const double* d_actions, int act_dim, int skip_frames, unsigned long long* h_out, size_t n) {
  MJRL_ENTER(e);
  if (n != 3 * (size_t)e->n_env) MJRL_FAIL(e, 4, "step_timeline: expected room for %zu counters, got %zu", 3 * (size_t)e->n_env, n);
  if (skip_frames != 1) MJRL_FAIL(e, 4, "step_timeline: one frame per step only");
  unsigned long long* d = nullptr;
  MJRL_HIP(e, hipMalloc(&d, sizeof(unsigned long long) * n));
  int rc = launch_step(e, d_actions, act_dim, skip_frames, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, d);
  if (!rc) {
    hipError_t he = hipStreamSynchronize(e->stream);
    if (he == hipSuccess) he = hipMemcpy(h_out, d, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { e->err = hipGetErrorString(he); rc = 100 + (int)he; }
  }
  hipFree(d);
  return rc;
}

int mjrl_query(mjrl_env* e, const char* name, double* h_out, size_t nbytes) {
  MJRL_ENTER(e);
  const DevModel& m = e->hm;
  // The frames of the last forward pass: kept by the step kernel when the query cache is on (what the reference's
  // data.xipos / data.contact hold after mj_step); otherwise, or after a state write, one forward-only launch.
  if (!e->frames) if (int rc = mjrl_set_query_cache(e, 1)) return rc;
  if (!e->frames_valid) {
    // (a pure query: the forward pass fills the frame cache and refreshes sensordata, the warm start stays)
    if (int rc = launch_step(e, nullptr, 0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 2, nullptr, nullptr)) return rc;
  }
  MJRL_HIP(e, hipStreamSynchronize(e->stream));
  const int fd = mj::frame_doubles(m);
  std::vector<double> img((size_t)e->n_env * fd);
  MJRL_HIP(e, hipMemcpy(img.data(), e->frames, sizeof(double) * img.size(), hipMemcpyDeviceToHost));
  const int o_xpos = 0, o_xquat = 3 * m.nbody, o_gpos = 7 * m.nbody, o_gquat = 7 * m.nbody + 3 * m.ngeom,
            o_ncon = 7 * m.nbody + 7 * m.ngeom, o_con = o_ncon + 1;
  size_t per = 0;
  int off = 0;
  enum { PLAIN, XIPOS, GMAT, WARN } kind = PLAIN;
  if (!strcmp(name, "xpos")) { per = 3 * m.nbody; off = o_xpos; }
  else if (!strcmp(name, "xquat")) { per = 4 * m.nbody; off = o_xquat; }
  else if (!strcmp(name, "geom_xpos")) { per = 3 * m.ngeom; off = o_gpos; }
  else if (!strcmp(name, "geom_xmat")) { per = 9 * m.ngeom; kind = GMAT; }
  else if (!strcmp(name, "xipos")) { per = 3 * m.nbody; kind = XIPOS; }
  else if (!strcmp(name, "ncon")) { per = 1; off = o_ncon; }
  else if (!strcmp(name, "contact_geom")) { per = 2 * m.nconmax; off = o_con; }
  else if (!strcmp(name, "warn")) { per = 1; kind = WARN; }
  else MJRL_FAIL(e, 4, "query: unknown quantity '%s'", name);
  if (nbytes != sizeof(double) * per * e->n_env) MJRL_FAIL(e, 4, "query(%s): buffer holds %zu bytes, need %zu", name, nbytes, sizeof(double) * per * e->n_env);
  if (kind == WARN) {
    // cap-overflow flags live in the LDS image only: one debug forward pass
    if (int rc = ensure_dbg(e)) return rc;
    // (this pass runs at the current state, not at the one the cached frames and scene rows belong to: it leaves both alone)
    double* frames = e->frames;
    const bool keep = e->frames_valid, scene_on = e->scene_on, scene_valid = e->scene_valid;
    e->frames = nullptr; e->scene_on = false;
    int rc = launch_step(e, nullptr, 0, 1, nullptr, nullptr, nullptr, nullptr, e->dbg, 0, 2, nullptr, nullptr);
    e->frames = frames; e->frames_valid = keep; e->scene_on = scene_on; e->scene_valid = scene_valid;
    if (rc) return rc;
    MJRL_HIP(e, hipStreamSynchronize(e->stream));
    std::vector<double> lds((size_t)e->n_env * e->lay.total);
    MJRL_HIP(e, hipMemcpy(lds.data(), e->dbg, sizeof(double) * lds.size(), hipMemcpyDeviceToHost));
    for (int env = 0; env < e->n_env; env++)
      h_out[env] = ((const int*)(lds.data() + (size_t)env * e->lay.total + e->lay.ints))[mj::I_WARN];
    return 0;
  }
  auto quat_to_mat = [](const double* q, double* r) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    r[0] = w * w + x * x - y * y - z * z; r[1] = 2 * (x * y - w * z); r[2] = 2 * (x * z + w * y);
    r[3] = 2 * (x * y + w * z); r[4] = w * w - x * x + y * y - z * z; r[5] = 2 * (y * z - w * x);
    r[6] = 2 * (x * z - w * y); r[7] = 2 * (y * z + w * x); r[8] = w * w - x * x - y * y + z * z;
  };
  for (int env = 0; env < e->n_env; env++) {
    const double* F = img.data() + (size_t)env * fd;
    double* o = h_out + (size_t)env * per;
    if (kind == PLAIN) memcpy(o, F + off, sizeof(double) * per);
    else if (kind == GMAT) {
      for (int g = 0; g < m.ngeom; g++) quat_to_mat(F + o_gquat + 4 * g, o + 9 * g);
    } else {
      for (int b = 0; b < m.nbody; b++) {
        double r[9];
        quat_to_mat(F + o_xquat + 4 * b, r);
        const double* ip = m.body_ipos + 3 * b;
        for (int k = 0; k < 3; k++)
          o[3 * b + k] = F[o_xpos + 3 * b + k] + r[3 * k] * ip[0] + r[3 * k + 1] * ip[1] + r[3 * k + 2] * ip[2];
      }
    }
  }
  return 0;
}

}  // extern "C"
