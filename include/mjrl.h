/* C-ABI of the MI355X batched env stepper (libmjrl_hip.so).
 *
 * This is the drop-in boundary for the reference's step path.  The reference is pure Python and
 * crosses into native code through the `mujoco` bindings; the entry points below are what a
 * ctypes binding for that path needs, each citing the reference call it
 * replaces.  Plain pointers and sizes only; no torch / numpy types.
 *
 * Threading: a handle is NOT thread-safe; use one handle per GPU (one process per GPU).
 * Ownership: the library owns all device buffers and the uploaded model; callers own every buffer
 * they pass.  Every entry returns 0 on success, non-zero on error; mjrl_last_error() explains.
 * Pointers named d_* are DEVICE pointers (HBM), h_* are host pointers.
 */
#ifndef MJRL_H
#define MJRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mjrl_env mjrl_env;

/* library / blob-layout version string */
const char* mjrl_version(void);

/* last error of a handle; pass NULL for the error of a failed mjrl_create */
const char* mjrl_last_error(const mjrl_env* env);

/* Upload a compiled model (blob.py layout) and allocate state for n_env copies on device_id.
 * Replaces: mj.MjModel.from_xml_path + mj.MjData (mujoco_parent.py:126-127), once per copy.
 * All copies start at qpos0 with zero velocity (as after mj_resetData, mujoco_parent.py:349).
 * flags bit 0: turn off the longest-first dispatch (copies are then stepped by workgroup id == copy id).
 * A batch of at most one wave per SIMD (n_env <= 4 x compute units) gets the step kernel's build for such batches
 * (mjrl_size "few": solver forms that use the whole register file, no dispatch order).  The rule assumes that the handle
 * has the device to itself: flags bit 1 = other handles step on this device at the same time, use the full-batch build
 * whatever n_env is; flags bit 2 = use the few-copies build whatever n_env is. */
int mjrl_create(const void* blob, size_t nbytes, int n_env, int device_id, unsigned flags, mjrl_env** out);
void mjrl_destroy(mjrl_env* env);

/* Launch on a caller-provided HIP stream (hipStream_t passed as void*); NULL = the handle's own. */
int mjrl_set_stream(mjrl_env* env, void* hip_stream);
/* Wait for everything queued on the handle's stream. */
int mjrl_sync(mjrl_env* env);

/* Observation gather table.  Replaces the table built by get_observation_space_mujoco
 * (mujoco_parent.py:233-272) and consumed by get_observations (mujoco_parent.py:380-392):
 * for each agent, sensordata addresses, then qpos indices, then qvel indices; counts are per agent,
 * index arrays are the per-agent lists concatenated. */
int mjrl_set_gather_tables(mjrl_env* env, int n_agent, const int32_t* n_sensor, const int32_t* sensor_idx,
                           const int32_t* n_qpos, const int32_t* qpos_idx, const int32_t* n_qvel,
                           const int32_t* qvel_idx);
/* Action scatter table.  Replaces agents_action_index (mujoco_parent.py:274-314) as used by apply_action
 * (mujoco_parent.py:323-332): mode 0 writes ctrl[idx], mode 1 overwrites qvel[idx] (freeJoint).
 * Action slots beyond n_idx[agent] are not routed to the physics (dynamics-plugin slots). */
int mjrl_set_scatter_tables(mjrl_env* env, int n_agent, int mode, const int32_t* n_idx, const int32_t* idx);
int mjrl_set_max_steps(mjrl_env* env, int max_steps);   /* truncation horizon, mujoco_rl.py:59,412 */

/* Fused on-device plugin vocabulary ("GPU fast path"): a short program run by the step kernel after the
 * observation gather, in list order, agent-minor, strictly sequentially -- the order of the reference's plugin loop
 * (environmentDynamics mujoco_rl.py:215-241, rewardFunctions :276-277, doneFunctions :281-286).  It replaces those
 * host loops for plugins that fit the vocabulary; anything else stays a host plugin.
 *   prog_i [n_op][8] = {kind, i1..i7}, prog_f [n_op][4]:
 *   kind 1 LANGUAGE      i1 action slot, i2 store slot, i3 extra-obs index   (README.md:109-136, 4-tuple form)
 *   kind 2 DIST_REWARD   i1 target kind (0 body xipos, 1 geom xpos), i2 target id, i3 store slot or -1, i4 mode; f0 scale
 *   kind 3 DIST_DONE     i1 target kind, i2 target id; f0 threshold
 *          (kinds 2, 3) target kind 2 = the agent's current target: i2 tag, i5 store slot holding its index in the tag's list
 *   kind 4 TARGET        i1 tag, i2 store slot of the current target's index, i3 store slot of the inventory or -1,
 *                        i4 extra-obs index, i5 store slot of the distance or -1; f0 threshold, f1 reward, f2 seed
 *                        (Testing/EnvironmentDynamic.py:17-32 target switching; Testing/Pick_Up_Dynamic.py:15-41 with the
 *                        inventory toggle): obs = the current target's position (3) [+ inventory]
 * n_slot doubles of device data store per (env, agent) (NaN = key absent; cleared by mjrl_reset, mujoco_rl.py:312),
 * n_extra_obs observation values appended after each agent's physical observation (obs_dim grows by it).
 * Must be called after mjrl_set_gather_tables. */
int mjrl_set_program(mjrl_env* env, int n_op, const int32_t* prog_i, const double* prog_f, int n_slot,
                     int n_extra_obs, const int32_t* agent_body);

/* Object tags of the level's info JSON as device tables (mujoco_rl.py:93-112 __instantiateJson, :355-378 filter_by_tag):
 * tag t names tag_num[t] objects, listed in filter_by_tag's order in tag_ref (concatenated over the tags); an entry is
 * kind << 16 | id with kind 0 = body (position xipos), 1 = geom (position xpos) -- the way get_data / distance resolve a
 * name (mujoco_parent.py:404-416, 440-446).  Ops of the fused program refer to a tag by its index.  Call before
 * mjrl_set_program. */
int mjrl_set_tag_tables(mjrl_env* env, int n_tag, const int32_t* tag_num, const int32_t* tag_ref);
/* Global id of copy 0 (a shard of a larger batch): on-device random choices (OP_TARGET, level variants) are keyed on
 * the global copy id, so a copy's episode does not depend on how the batch is sharded. */
int mjrl_set_env_base(mjrl_env* env, int first_env_id);
/* Per-copy level variants: an xmlPath list whose levels differ in geom colours only (Testing/levels/Model2-10.xml) is
 * one model plus n_variant colour tables rgba [n_variant][ngeom][4].  Every reset of a copy (mjrl_reset*, in-launch
 * reset) draws its variant like the reference's random.choice(xml_paths) at reset (mujoco_parent.py:351-356), keyed on
 * (seed, global copy id, episode count); the cameras render a copy in its variant's colours.  Fields "variant" and
 * "episode" (int32 [n_env]) are readable and writable through mjrl_get_field / mjrl_set_field.  n_variant 0 turns it off. */
int mjrl_set_variants(mjrl_env* env, int n_variant, const double* rgba, unsigned long long seed);

/* Sizes a caller needs to allocate buffers: "nq","nv","nu","nbody","ngeom","nsensordata","obs_dim",
 * "n_agent","n_env","lds_doubles","ncon_stride", ...; "few": 1 when the batch leaves every SIMD of the device at most one
 * wave (n_env <= 4 x the CUs) -- the step kernel then uses solver forms that need the whole register file, and a
 * model-specialised kernel for such a handle is built with -DMJRL_FEW=1 (kernel_cache.code_object(blob, few=True)); -1 for
 * an unknown name. */
int mjrl_size(const mjrl_env* env, const char* name);

/* Reset copies: qpos0 / zero velocity / zero ctrl / timestep 0 / empty data store, followed by mj_forward.
 * Replaces mj_resetData + mj_forward (mujoco_parent.py:349-350); mask is a host array [n_env] of 0/1, NULL = all.
 * Every copy resets to the same state, so what mj_forward leaves there (sensordata, qacc_warmstart) is computed once
 * per model at mjrl_create and copied; copies outside the mask are not touched.
 * If d_obs is non-NULL the post-reset observations of the selected copies (sensordata | qpos | qvel,
 * mujoco_rl.py:314; slots owned by fused dynamics: 0) are written to their rows of d_obs [n_env][n_agent][obs_dim]. */
int mjrl_reset(mjrl_env* env, const uint8_t* h_mask, double* d_obs);
/* The same with the mask in device memory (e.g. the term | trunc bytes a step wrote): asynchronous, no host copy. */
int mjrl_reset_device(mjrl_env* env, const uint8_t* d_mask, double* d_obs);
/* In-launch reset for sampling loops (`if done: env.reset()` then `env.step(a)`, fps_benchmark.py:33-38): while a mask
 * is set, every step launch first resets the copies whose byte is non-zero at that moment -- exactly as mjrl_reset would
 * -- and then steps them, in the same launch.  The mask [n_env] is caller-owned device memory, read by each launch
 * (typically the done flags of the previous step); NULL turns it off.  The reset observation is not produced: the
 * step returns the observation after the first step of the new episode. */
int mjrl_set_step_reset_mask(mjrl_env* env, const uint8_t* d_mask);
/* I/O layout of the step entries for single-agent consumers (the vector-env adapter over wrappers.py:12-82's use
 * case: one driven agent).  agent >= 0: `actions` holds that agent's row only, [n_env][act_dim] -- the other agents act
 * with 0 --, and `obs` receives only its row, [n_env][obs_dim]; obs_f32 != 0: observations are written as float (half
 * the bytes a host caller pulls over PCIe through the pinned buffers).  reward / term / trunc stay [n_env][n_agent].
 * Holds for mjrl_step_device, mjrl_step_pinned (the pinned buffers keep their size; the layout's rows come first) and the
 * reset observations of mjrl_reset / mjrl_reset_device; mjrl_step_host refuses a non-default layout.  agent -1,
 * obs_f32 0 restores the layouts documented at those entries. */
int mjrl_set_io_layout(mjrl_env* env, int agent, int obs_f32);

/* (round 3) A mask byte of 2 resets the copy WITHOUT stepping it: no physics frame runs for it in that step, its rows of
 * the step's outputs hold the reset observation (what reset() returns, mujoco_rl.py:314; slots owned by fused dynamics and
 * camera latents: 0 / the encoding of the reset state's image), reward 0, termination and truncation clear; its step
 * counter is 0 afterwards.
 *
 * Autoreset kept on the device -- what a vector-env adapter (Gymnasium VectorEnv / SB3 VecEnv over
 * MuJoCo_Gym/wrappers.py:12-82's single-agent shim) needs: every step records per copy whether the copy's episode ended in
 * it (a termination or truncation flag of any agent), and the NEXT step resets the copies so flagged --
 *   mode 1: without stepping them (a mask byte of 2): Gymnasium's next-step autoreset, the step after an episode's end
 *           returns the first observation of the new episode;
 *   mode 2: and then steps them (a mask byte of 1): `env.reset(); env.step(a)` of the reference's sampling loops;
 *   mode 0: off.
 * No mask travels between host and device, on any of the step entry points.  mjrl_reset of a copy clears its flag; an
 * explicit step-reset mask byte wins over it. */
int mjrl_set_autoreset(mjrl_env* env, int mode);

/* One step() of every copy: scatter actions, skip_frames physics steps, gather observations.
 * Replaces apply_action + mj_step loop (mujoco_parent.py:316-336) and get_observations (:380-392),
 * plus the zero-initialised rewards / terminations and the truncation flag of MuJoCoRL.step
 * (mujoco_rl.py:262-263, 279).  All pointers are device pointers and may be NULL except d_actions
 * when a scatter table is set.  Asynchronous on the handle's stream.
 *   d_actions [n_env][n_agent][act_dim]   d_obs   [n_env][n_agent][obs_dim]
 *   d_reward  [n_env][n_agent]            d_term / d_trunc [n_env][n_agent] (bytes) */
int mjrl_step_device(mjrl_env* env, const double* d_actions, int act_dim, int skip_frames, double* d_obs,
                     double* d_reward, uint8_t* d_term, uint8_t* d_trunc);
/* Host buffers without copies: pinned host memory owned by the handle and mapped into the device's address space.
 * mjrl_host_buffers returns (allocating on the first call, and again when act_dim or the observation width grew) the
 * buffers mjrl_step_pinned works on: the caller writes actions [n_env][n_agent][act_dim] into *h_actions, calls
 * mjrl_step_pinned (synchronous), and reads obs / reward / term / trunc in place -- the kernel reads the action rows
 * and writes the result rows over PCIe itself, the observations under the rest of the launch, so a step costs the launch
 * and not the launch plus five copies.  The buffers stay valid until the handle is destroyed or a later
 * mjrl_host_buffers call has to grow them; every step overwrites them.  What mjrl_step_host does for caller-owned
 * (pageable) arrays -- the shape of the reference's numpy API (mujoco_rl.py:243-289 returns fresh dicts every step). */
int mjrl_host_buffers(mjrl_env* env, int act_dim, double** h_actions, double** h_obs, double** h_reward,
                      uint8_t** h_term, uint8_t** h_trunc);
int mjrl_step_pinned(mjrl_env* env, int act_dim, int skip_frames);
/* Same with host buffers (PCIe copies in and out, synchronous). */
int mjrl_step_host(mjrl_env* env, const double* h_actions, int act_dim, int skip_frames, double* h_obs,
                   double* h_reward, uint8_t* h_term, uint8_t* h_trunc);

/* State access for parity tests and host-side plugins (synchronous, host buffers [n_env][n]).
 * Fields: "qpos","qvel","ctrl","qacc_warmstart","sensordata","timestep"(int32),"store" [n_env][n_agent][n_slot],
 * "solver_stats"(int32, read-only) [n_env][4] = data.ncon, data.nefc, solver sweeps, cap-warning bits (1 nconmax,
 * 2 njmax) of each copy's last physics frame. */
int mjrl_get_field(mjrl_env* env, const char* name, void* h_out, size_t nbytes);
int mjrl_set_field(mjrl_env* env, const char* name, const void* h_in, size_t nbytes);

/* Keep the frames of every step's forward pass in HBM for mjrl_query (on by default once mjrl_query is used; turn
 * it on up front when host plugins will query, so that the first query after a step is served from that step). */
int mjrl_set_query_cache(mjrl_env* env, int enabled);

/* Derived quantities for host-side plugins (get_data / distance / collision, mujoco_parent.py:394-478), as the
 * reference's MjData holds them after mj_step: the frames of the LAST FORWARD PASS (inside the last step, or of the
 * forward pass after a reset / state write):
 *   "xpos" [n_env][nbody][3]  "xquat" [n_env][nbody][4]  "xipos" [n_env][nbody][3]
 *   "geom_xpos" [n_env][ngeom][3]  "geom_xmat" [n_env][ngeom][9]
 *   "ncon" [n_env] (as double)  "contact_geom" [n_env][nconmax][2] (as double, -1 padded). */
int mjrl_query(mjrl_env* env, const char* name, double* h_out, size_t nbytes);

/* Agent cameras (get_camera_data, mujoco_parent.py:540-555): every fixed camera of the level, ray cast;
 * rgb [n_env][ncam][height][width][3] uint8, rows bottom-up as glReadPixels returns them (mujoco_parent.py:571, 538).
 * The image model is OpenGL's fixed-function lighting equation with the parameters MuJoCo documents (headlight, the
 * level's <light>s, materials) with the lights' shadows as shadow rays (castshadow); textures, the skybox and
 * anti-aliasing are not drawn.
 *
 * WHICH frames are drawn: the reference calls mjv_updateScene(model, data, ...) (mujoco_parent.py:533), which reads the
 * geom / camera / light frames out of MjData as the last forward pass left them -- after mj_step those are one
 * integration older than qpos.  mjrl_set_scene_cache(env, 1) gives exactly that: every step (its last physics frame),
 * forward pass and reset then leaves each copy's frames in a scene row, and the render entries draw the rows.  Turn it on
 * before stepping (mjrl_set_camera_obs does).  With the cache off, or after a state was written by hand
 * (mjrl_set_field), the frames are computed from the CURRENT qpos instead (one extra kernel per call). */
int mjrl_set_scene_cache(mjrl_env* env, int enabled);
int mjrl_render_device(mjrl_env* env, int width, int height, uint8_t* d_rgb);
int mjrl_render_host(mjrl_env* env, int width, int height, uint8_t* h_rgb);

/* Encoder of the reference's vision autoencoder (vision/autoencoder.py:12-18) on the matrix cores:
 *   Conv2D(32, 3x3, relu, stride 2, same) -> Conv2D(64, 3x3, relu, stride 2, same) -> Flatten -> Dense(latent_dim[, relu])
 * on 64x64x3 uint8 images scaled by 1/255 (vision/train.py:26).  Weights as Keras stores them, float32, channels last:
 *   w1 [3][3][3][32]  b1 [32]  w2 [3][3][32][64]  b2 [64]  wd [16384][latent_dim]  bd [latent_dim]   (host pointers).
 * Arithmetic: bf16 operands (weights and the activations between the layers are rounded to bf16), fp32 accumulation. */
int mjrl_encoder_load(mjrl_env* env, int latent_dim, int relu_latent, const float* w1, const float* b1, const float* w2,
                      const float* b2, const float* wd, const float* bd);
/* Encode n_img images [n_img][64][64][3] (uint8, e.g. the output of mjrl_render_device at 64 x 64) to
 * float32 latents [n_img][latent_dim]. */
int mjrl_encode_device(mjrl_env* env, const uint8_t* d_rgb, int n_img, float* d_latent);
int mjrl_encode_host(mjrl_env* env, const uint8_t* h_rgb, int n_img, float* h_latent);
/* Camera latents inside the observation: agent a's observation row grows by latent_dim values -- the encoding of the
 * 64x64 image of camera agent_cam[a] (-1: the agent has no camera; its slots stay 0) -- placed after the physical part
 * and the fused program's values.  Every mjrl_step_* call that returns observations then renders and encodes behind
 * the step on the same stream (what a vision policy does with get_camera_data + the autoencoder, mujoco_parent.py:540-555,
 * vision/autoencoder.py).  Needs mjrl_encoder_load and the gather tables; n_agent 0 turns it off.  obs_dim changes. */
int mjrl_set_camera_obs(mjrl_env* env, int n_agent, const int32_t* agent_cam);

/* Debug: step once like mjrl_step_device and also dump every copy's LDS image
 * ([n_env][lds_doubles] doubles; stage 0 = end of the forward pass, 1 = after the row build).
 * mjrl_lds_offset gives the offset of a named region inside one image. */
int mjrl_step_debug(mjrl_env* env, const double* d_actions, int act_dim, int skip_frames, int stage,
                    double* h_dump, size_t nbytes);
int mjrl_lds_offset(const mjrl_env* env, const char* region);
/* Diagnostic: one step with per-stage wave-clock stamps; h_cycles[n] receives, per stage, the cycles summed over all
 * env copies.  Stage order: load kin com crb factor geom collide vel smooth rows project pgs sensors euler store
 * pgs_warm pgs_lists pgs_sweeps rows_limits rows_addr pgs_setup tail.  "tail" is what follows the state store (work-bucket
 * filing, truncation flags, fused plugin ops); the six before it are parts of "rows" and "pgs"
 * (which then hold the remainder: the row build proper; the solver's return to joint space); for the register solver
 * pgs_setup is the AR build and pgs_sweeps the sweeps alone.  n must be 22. */
int mjrl_step_profile(mjrl_env* env, const double* d_actions, int act_dim, int skip_frames,
                      unsigned long long* h_cycles, int n);

/* Diagnostic: one launch of the step kernel in which every wave ends right after stage number `stop_after` (1-based, in
 * the stage order of mjrl_step_profile; 1..15 are the stages of the step proper) and nothing is written back: the copies'
 * state is untouched.  Hardware counters (rocprofv3 --pmc) of launches cut at successive stages give each stage's
 * instruction mix by difference (tools/stage_mix.py).  The cuts exist only in a specialised kernel built with
 * -DMJRL_DIAG -DMJRL_STAGE_CUT (MJRL_SPEC_FLAGS); any other kernel leaves such a launch at once.  (Round 3: the production
 * kernels carry no diagnostics at all; mjrl_step_debug / _profile / _timeline / _truncated and the LDS read-back of
 * mjrl_query launch the diagnostic build -- mjrl_step_kernel_diag of the library, or a specialised kernel built with
 * -DMJRL_DIAG.  Same source, same arithmetic, same bits.) */
int mjrl_step_truncated(mjrl_env* env, int stop_after);

/* Diagnostic: one step whose waves record when they ran.  h_out[3*w + 0..2] = start and end of workgroup w's wave on
 * the device's constant 100 MHz clock and the env copy it stepped (w is the dispatch order, which the longest-first
 * scheduling decouples from the copy index).  n must be 3 * n_env; skip_frames must be 1. */
int mjrl_step_timeline(mjrl_env* env, const double* d_actions, int act_dim, int skip_frames,
                       unsigned long long* h_out, size_t n);

/* Cap overflows since creation (or the last clearing call): h_counts[0] = physics frames, summed over env copies, whose
 * contact list was cut at nconmax; h_counts[1] = frames whose constraint rows were cut at njmax.  The counterpart of
 * MuJoCo's mjWARN_CONTACTFULL / mjWARN_CNSTRFULL counters in data.warning (the reference never reads them; mujoco
 * 2.3.3 prints a warning).  A non-zero count means nconmax / njmax (config keys of the same names) are too small for
 * the level.  clear != 0 zeroes the counters after reading. */
int mjrl_cap_overflows(mjrl_env* env, unsigned long long* h_counts, int clear);

/* Attach a model-specialised build of the step kernel: a gfx950 code object made from csrc/mjrl_spec_kernel.hip with
 * the model's sizes as compile-time constants (kernel_cache.py drives hipcc --genco and caches the result next to the
 * library).  The code object carries the sizes it was built for; a mismatch with this batch's model is an error and
 * leaves the generic kernel in place.  path == NULL detaches.  The arithmetic is the generic kernel's, operation for
 * operation, so results are bit-identical; only address computation, loop structure and register use change.
 * (No counterpart in the reference: MuJoCo's C step is shape-generic; this is the GPU build's stand-in for a JIT.) */
int mjrl_load_kernel(mjrl_env* env, const char* path);

#ifdef __cplusplus
}
#endif

#endif
