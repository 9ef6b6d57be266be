"""Round-3 tests on the MI355X, through the C-ABI: the fixes of the round-2 review (pinned host buffers re-sized when the
agent count changes, a copy's episodes draw different targets, latent slots of an agent without a camera read 0,
``MuJoCoRL.close`` releases the device state under a ParallelEnv base, truncation flags per copy) and the production /
diagnostic kernel split (the diagnostic entry points still work and give the production kernel's bits)."""
import numpy as np
import pytest

from mjrl_amd import _capi, blob, levels, mjcf
from mjrl_amd.mujoco_rl import MuJoCoRL
from oracle.oracle import OracleEnv

pytestmark = pytest.mark.gpu

AGENTS = ["sender", "receiver"]
INFO_JSON = __file__.rsplit("/", 1)[0] + "/golden/two_agent_info.json"


def test_pinned_buffers_follow_the_agent_count():
    """mjrl_host_buffers before the gather / scatter tables exist sees no agents and sizes the reward / flag buffers for
    one element per copy; a second call after the tables are set must re-size them (the kernel writes n_env x n_agent
    elements through the mapped pointers), and step_pinned must refuse a stale set."""
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    h = _capi.Handle(blob.pack(model), 40)
    h.reset()
    early = h.host_buffers(8)
    assert early[2].shape == (40, 1)                       # no agents yet
    env = MuJoCoRL.tables_only(levels.level_path("two_agent.xml"))
    sens, qp, qv, scat = [], [], [], []
    for a in AGENTS:
        env.get_observation_space_mujoco(a)
        env.get_action_space_mujoco(a)
        idx = env.agents_observation_index[a]
        sens.append(idx["sensors"]); qp.append(idx["qpos"]); qv.append(idx["qvel"])
        scat.append(env.agents_action_index[a])
    h.set_gather_tables(sens, qp, qv)
    h.set_scatter_tables(scat, 0)
    with pytest.raises(Exception, match="host_buffers"):
        h.step_pinned(8, 1)                               # the early buffers are too small for two agents
    act, obs, rew, term, trunc = h.host_buffers(8)
    assert rew.shape == (40, 2) and term.shape == (40, 2) and obs.shape == (40, 2, 59)
    rew[:] = 7.0; term[:] = 9; trunc[:] = 9
    rng = np.random.default_rng(0)
    act[:] = rng.uniform(-1, 1, act.shape)
    h.step_pinned(8, 1)
    assert not rew.any() and not term.any() and not trunc.any()          # all 80 elements were written by the kernel
    ora = OracleEnv(blob.pack(model))
    ora.reset()
    for k in range(2):
        ora.ctrl[scat[k]] = act[39, k]
    ora.step()
    assert np.allclose(obs[39, 0, 1:31], ora.qpos, atol=1e-12)
    h.close()


def test_target_draws_differ_between_episodes_and_match_the_host_plugin():
    """The key of an on-device random choice holds the copy's episode count (kept whether or not level variants are on):
    the first target of a copy's second episode is not tied to the first episode's, the host plugin of the same name
    draws the same numbers over two episodes, and an in-launch reset counts an episode like mjrl_reset does."""
    import torch
    from mjrl_amd.dynamics import PickUpDynamic, episode_key, mix64, pick_of

    class Pick(PickUpDynamic):
        threshold, seed = 0.01, 7          # (never reached: what is under test are the episodes' first draws)

    def make_env(fused, n=64):
        return MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "infoJson": INFO_JSON, "agents": AGENTS, "numEnvs": n,
                         "environmentDynamics": [Pick], "fusedPlugins": fused, "firstEnvId": 100, "maxSteps": 6})
    fused, host = make_env(True), make_env(False)
    firsts = []
    rng = np.random.default_rng(1)
    for episode in range(2):
        np.random.seed(0); fused.reset()
        np.random.seed(0); host.reset()
        assert np.array_equal(fused.episode, np.full(64, episode + 1)) and np.array_equal(host.episode, fused.episode)
        for step in range(5):
            action = {a: rng.uniform(-1, 1, (64, 8)) for a in AGENTS}
            f, h = fused.step(action), host.step(action)
            for a in AGENTS:
                assert np.allclose(f[0][a], h[0][a], atol=1e-12), (episode, step, a)
                assert np.allclose(f[1][a], h[1][a], atol=1e-10)
            if step == 0:
                firsts.append(fused.device_store["sender"]["current_target"].copy())
                assert (episode_key(host) == ((np.uint64(episode + 1) << np.uint64(32)) | np.uint64(1))).all()
                want = pick_of(mix64(7, 100 + np.arange(64), 0, np.full(64, np.uint64(episode + 1) << np.uint64(32)), 0), 3)
                assert np.array_equal(firsts[-1], want.astype(np.float64))
    assert (firsts[0] != firsts[1]).any()                 # fresh choices each episode (random.randint in the reference)
    # in-launch reset: the flagged copies start episode 3 inside the step launch and draw that episode's first target
    mask = np.zeros(64, np.uint8); mask[::3] = 1
    d_mask = torch.from_numpy(mask).cuda()
    fused._handle.set_step_reset_mask(d_mask.data_ptr())
    act = torch.zeros((64, 2, 8), dtype=torch.float64, device="cuda")
    fused.step_batched(act)
    torch.cuda.synchronize()
    fused._handle.set_step_reset_mask(None)
    ep = fused._handle.get_field("episode")
    assert np.array_equal(ep, 2 + mask)
    cur = fused.device_store["sender"]["current_target"]
    want3 = pick_of(mix64(7, 100 + np.arange(64), 0, (np.uint64(3) << np.uint64(32)), 0), 3).astype(np.float64)
    assert np.array_equal(cur[mask == 1], want3[mask == 1])
    fused.close(); host.close()


def test_latent_slots_of_an_agent_without_a_camera_read_zero():
    """include/mjrl.h: with mjrl_set_camera_obs, the latent slots of an agent whose camera id is -1 read 0 every step
    (nobody writes them: they used to hold whatever the caller's buffer held)."""
    import torch
    from tests.test_gpu_encoder import make_weights
    latent = 16
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 8, "agentCameras": True,
                    "cameraEncoder": {"weights": make_weights(latent, seed=2), "relu": True}})
    env._handle.set_camera_obs([0, -1])                   # the receiver loses its camera
    env.reset_batched()
    obs = torch.full((8, 2, 59 + latent), 123.0, dtype=torch.float64, device="cuda")
    act = torch.zeros((8, 2, 8), dtype=torch.float64, device="cuda")
    for _ in range(3):
        env.step_batched(act, obs)
    torch.cuda.synchronize()
    o = obs.cpu().numpy()
    assert not o[:, 1, 59:].any()                         # zeros, not the 123s of the caller's buffer
    assert np.abs(o[:, 0, 59:]).max() > 0 and not (o[:, 0, 59:] == 123.0).any()
    env.close()
    assert model.ncam == 2


def test_close_releases_the_device_state_under_a_parallel_env_base():
    """Where pettingzoo is installed ``ParallelEnv.close`` (a no-op) comes first in the MRO: ``MuJoCoRL.close`` must still
    destroy the handle and drop the views of its pinned buffers."""
    class StubParallelEnv:
        def close(self):
            pass

    class Env(StubParallelEnv, MuJoCoRL):
        pass

    env = Env({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 4})
    env.reset_batched()
    env.step_batched(np.zeros((4, 2, 8)))
    assert env._pinned is not None and env._handle is not None
    MuJoCoRL.close(env)
    assert env._handle is None and env._pinned is None


def test_truncation_flags_of_the_dict_api_are_per_copy():
    """A masked reset_batched leaves the copies at different points of their episodes; the dict API's truncation flags
    are the kernel's (each copy's own step counter), the same as step_batched returns."""
    from mjrl_amd.dynamics import Language
    cfg = {"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 4, "maxSteps": 3,
           "environmentDynamics": [Language]}
    env = MuJoCoRL(cfg)
    env.reset()
    zero = {a: np.zeros((4, 9)) for a in AGENTS}
    for _ in range(2):
        env.step(zero)
    env.reset_batched(mask=np.array([0, 1, 0, 1], np.uint8))
    out = [env.step(zero)[3] for _ in range(3)]
    # copies 0 and 2 are at steps 2, 3, 4 of their episode (truncated from step 3 on), copies 1 and 3 at 0, 1, 2
    assert out[0]["sender"].tolist() == [False, False, False, False]
    assert out[1]["sender"].tolist() == [True, False, True, False]
    assert out[2]["__all__"].tolist() == [True, False, True, False]
    env.close()


def test_diagnostic_launches_give_the_production_bits():
    """The production kernels carry no diagnostics; mjrl_step_debug / _profile / _timeline launch the diagnostic build.
    A batch stepped through step_debug (LDS dump) and one stepped through step_device end in the same state, bit for
    bit, and the dump shows the solver's work."""
    import torch
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    packed = blob.pack(model)
    a, b = _capi.Handle(packed, 32), _capi.Handle(packed, 32)
    assert a.kernel == "specialised"
    for h in (a, b):
        h.reset()
        h.set_scatter_tables([list(range(model.nu))], 0)
    rng = np.random.default_rng(5)
    q = np.tile(model.qpos0, (32, 1)); q[:, 2] = rng.uniform(0.1, 0.6, 32); q[:, 17] = rng.uniform(0.1, 0.6, 32)
    a.set_field("qpos", q); b.set_field("qpos", q)
    ioff = a.lds_offset("ints")
    for t in range(30):
        ctrl = torch.from_numpy(rng.uniform(-1, 1, (32, model.nu))).cuda()
        img = a.step_debug(ctrl.data_ptr(), model.nu, 1, 0)
        b.step_device(ctrl.data_ptr(), model.nu, 1)
    for f in ("qpos", "qvel", "qacc_warmstart"):
        assert np.array_equal(a.get_field(f), b.get_field(f)), f
    ints = img[:, ioff:ioff + 4].copy().view(np.int32)
    stats = b.get_field("solver_stats")
    assert np.array_equal(ints[:, 0], stats[:, 0]) and np.array_equal(ints[:, 1], stats[:, 1]) and stats[:, 1].max() > 8
    prof = b.step_profile()
    assert prof["pgs"] + prof["pgs_sweeps"] > 0 and prof["kin"] > 0
    tl = b.step_timeline()
    assert (tl[:, 1] > tl[:, 0]).all() and sorted(tl[:, 2].tolist()) == list(range(32))
    a.close(); b.close()


# --------------------------------------------------------------------------- vector-env adapter on the fast path (8f rank 1)
def _oracle_obs(o):
    return np.concatenate([o.sensordata[[0]], o.qpos, o.qvel])


@pytest.mark.parametrize("mode", ["next_step", "reset_then_step", "same_step"])
@pytest.mark.parametrize("path", ["numpy", "torch"])
def test_batched_vector_env_follows_the_oracle_through_autoresets(mode, path, few_build):
    """SURVEY 8f rank 1 (MuJoCo_Gym/wrappers.py:12-82 as a vector env): ``BatchedVectorEnv`` over ``mjrl_step_pinned``
    (numpy in / out) and ``mjrl_step_device`` (torch in / out), the autoreset kept by the kernel (mjrl_set_autoreset).  Every
    copy follows its own CPU oracle through three episodes; the three autoreset conventions return what their
    definitions say at the episode boundaries (call max_steps + 1 truncates, mujoco_rl.py:412)."""
    import torch
    from mjrl_amd.wrappers import BatchedVectorEnv
    n_env, horizon = 5, 6
    vec = BatchedVectorEnv(MuJoCoRL({"xmlPath": levels.level_path("single_agent.xml"), "agents": ["sender"],
                                     "numEnvs": n_env, "maxSteps": horizon}), autoreset=mode)
    assert vec.observation_space.shape == (n_env, 30) and vec.action_space.shape == (n_env, 8)
    obs, _ = vec.reset()
    env = vec.environment
    oras = [OracleEnv(env._blob) for _ in range(n_env)]
    first = np.stack([_oracle_obs(o) for o in oras])
    assert np.allclose(obs, first, atol=1e-12)
    rng = np.random.default_rng(6)
    idx = env.agents_action_index["sender"]
    host = lambda x: x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)
    t_in_episode, pending = 0, False           # oracle-side bookkeeping of the convention under test
    for step in range(1, 3 * horizon + 8):
        act = rng.uniform(-1, 1, (n_env, 8))
        out = vec.step(torch.from_numpy(act).cuda() if path == "torch" else act)
        obs, rew, term, trunc, info = host(out[0]), host(out[1]), host(out[2]), host(out[3]), out[4]
        assert not term.any() and not rew.any()
        if pending and mode == "next_step":
            # the copies are reset instead of stepped: the reset observation, flags clear, the action ignored
            for o in oras:
                o.reset()
            assert np.allclose(obs, first, atol=1e-12) and not trunc.any()
            pending, t_in_episode = False, 0
            continue
        if pending and mode == "reset_then_step":
            for o in oras:
                o.reset()
            pending, t_in_episode = False, 0
        for e, o in enumerate(oras):
            o.ctrl[idx] = act[e]
            o.step()
        expect = np.stack([_oracle_obs(o) for o in oras])
        t_in_episode += 1
        if t_in_episode == horizon + 1:          # the truncated call
            assert trunc.all()
            if mode == "same_step":
                assert np.allclose(host(info["final_observation"]), expect, atol=1e-9) and host(info["_final_observation"]).all()
                assert np.allclose(obs, first, atol=1e-12)
                for o in oras:
                    o.reset()
                t_in_episode = 0
            else:
                assert np.allclose(obs, expect, atol=1e-9)
                pending = True
        else:
            assert not trunc.any() and np.allclose(obs, expect, atol=1e-9), (step, t_in_episode)
    vec.close()


def test_vector_env_autoreset_is_per_copy_and_runs_the_fused_channel(few_build):
    """Copies that end their episodes at different steps are reset one by one (next-step convention) while the others
    keep stepping, on the 2-agent level with the fused Language channel driven through the adapter: the ended copy's
    row is the reset observation with the channel's slot at 0, its data store is empty, and it matches a fresh oracle
    from there on."""
    from mjrl_amd.dynamics import Language
    from mjrl_amd.wrappers import BatchedVectorEnv
    n_env = 6
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": n_env, "maxSteps": 5,
                    "environmentDynamics": [Language]})
    vec = BatchedVectorEnv(env, agent="sender")
    vec.reset()
    # stagger the copies: copy e starts e steps into its episode
    env._handle.set_field("timestep", np.arange(n_env, dtype=np.int32))
    rng = np.random.default_rng(0)
    ended_at = {}
    for step in range(12):
        act = rng.uniform(-1, 1, (n_env, 9)); act[:, 8] = 2.5
        obs, rew, term, trunc, _ = vec.step(act)
        obs, trunc = obs.copy(), trunc.copy()
        for e in range(n_env):
            if e in ended_at and ended_at[e] == step - 1:
                assert not trunc[e] and np.array_equal(obs[e, 1:31], env._compiled.qpos0) and obs[e, 59] == 0.0
            if trunc[e]:
                ended_at[e] = step
    assert len(ended_at) == n_env and len(set(ended_at.values())) > 1
    ts = env._handle.get_field("timestep")
    assert len(set(ts.tolist())) > 1                     # the copies are still at different points of their episodes
    vec.close()


@pytest.mark.parametrize("f32", [False, True])
def test_one_agent_io_layout_gives_the_default_layouts_rows(f32, few_build):
    """mjrl_set_io_layout (what the vector-env adapter runs on): with the driven agent's action row alone in the buffer
    and the other agent at 0, the kernel writes exactly the driven agent's row of the default layout -- the same bits as
    float64, the same values rounded to float with obs_f32 -- through the device entry and the pinned entry, with the
    fused Language channel's slot in the row, and through the reset observations."""
    import torch
    from mjrl_amd.dynamics import Language
    n_env, steps = 7, 40
    cfg = {"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": n_env, "maxSteps": 15,
           "environmentDynamics": [Language]}
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1, 1, (steps, n_env, 9)); acts[:, :, 8] = rng.integers(0, 3, (steps, n_env)) + 0.5
    for driven in (0, 1):
        ref, dev, pin = MuJoCoRL(cfg), MuJoCoRL(cfg), MuJoCoRL(cfg)
        for env in (ref, dev, pin):
            env.reset_batched()
            env._handle.set_autoreset(2)
        for env in (dev, pin):
            env._handle.set_io_layout(driven, f32)
        p_act, p_obs, p_rew, p_term, p_trunc = pin._handle.host_buffers(9)
        assert p_act.shape == (n_env, 9) and p_obs.shape == (n_env, 60) and p_obs.dtype == (np.float32 if f32 else np.float64)
        d_obs = torch.empty((n_env, 60), dtype=torch.float32 if f32 else torch.float64, device="cuda")
        d_rew = torch.empty((n_env, 2), dtype=torch.float64, device="cuda")
        d_term = torch.empty((n_env, 2), dtype=torch.uint8, device="cuda")
        d_trunc = torch.empty((n_env, 2), dtype=torch.uint8, device="cuda")
        for t in range(steps):
            full = np.zeros((n_env, 2, 9))
            full[:, driven] = acts[t]
            obs, rew, term, trunc = ref.step_batched(torch.from_numpy(full).cuda())
            want = obs[:, driven].cpu().numpy()
            d_act = torch.from_numpy(acts[t]).cuda()
            dev._handle.step_device(d_act.data_ptr(), 9, 1, d_obs.data_ptr(), d_rew.data_ptr(), d_term.data_ptr(), d_trunc.data_ptr())
            p_act[:] = acts[t]
            pin._handle.step_pinned(9, 1)
            for got in (d_obs.cpu().numpy(), p_obs):
                assert np.array_equal(got, want.astype(np.float32) if f32 else want), (driven, t)
            assert np.array_equal(d_trunc.cpu().numpy(), trunc.cpu().numpy()) and np.array_equal(p_trunc, trunc.cpu().numpy())
            assert np.array_equal(d_rew.cpu().numpy(), rew.cpu().numpy())
        assert trunc.cpu().numpy().any() or steps > 15                 # (episodes ended and restarted on the way)
        # reset observations in the same layout
        r_full = torch.empty((n_env, 2, 60), dtype=torch.float64, device="cuda")
        ref._handle.reset_device(None, r_full.data_ptr())
        dev._handle.reset_device(None, d_obs.data_ptr())
        want = r_full[:, driven].cpu().numpy()
        assert np.array_equal(d_obs.cpu().numpy(), want.astype(np.float32) if f32 else want)
        with pytest.raises(Exception, match="default layout"):
            dev._handle.step_host(np.zeros((n_env, 2, 9)), 1, np.zeros((n_env, 2, 60)))
        dev._handle.set_io_layout(-1, False)
        dev._handle.step_host(np.zeros((n_env, 2, 9)), 1, np.zeros((n_env, 2, 60)))
        for env in (ref, dev, pin):
            env.close()


def test_vector_env_results_are_the_callers_to_keep_unless_asked_otherwise():
    """copy=True (the default, Gymnasium's own convention): what step() returned is unchanged by later steps -- fresh numpy
    arrays, fresh torch tensors; copy=False hands out the buffers the next step overwrites.  float32 observations are the
    float64 ones rounded."""
    import torch
    from mjrl_amd.wrappers import BatchedVectorEnv
    n_env = 9
    make = lambda **kw: BatchedVectorEnv(MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS,
                                                   "numEnvs": n_env, "maxSteps": 50}), agent="receiver", **kw)
    rng = np.random.default_rng(1)
    acts = rng.uniform(-1, 1, (6, n_env, 8))
    keep, view, f32 = make(), make(copy=False), make(obs_dtype=np.float32)
    for v in (keep, view, f32):
        v.reset()
    held = []
    for t in range(6):
        o_keep = keep.step(acts[t])[0]
        o_view = view.step(acts[t])[0]
        o_f32 = f32.step(acts[t])[0]
        assert o_f32.dtype == np.float32 and np.array_equal(o_f32, o_keep.astype(np.float32))
        assert np.array_equal(o_keep, o_view)
        held.append((o_keep, o_keep.copy(), o_view))
    assert all(np.array_equal(a, b) for a, b, _ in held)              # copies stay what they were
    assert np.array_equal(held[0][2], held[-1][2]) and not np.array_equal(held[0][1], held[-1][1])   # views alias one buffer
    for v in (keep, view, f32):
        v.close()
    tk = make()
    tk.reset()
    outs, snap = [], []
    for t in range(6):
        outs.append(tk.step(torch.from_numpy(acts[t]).cuda())[0])
        snap.append(outs[-1].clone())
    assert outs[0].dtype == torch.float64 and all(torch.equal(a, b) for a, b in zip(outs, snap))     # every one still its own
    assert len({o.data_ptr() for o in outs}) == 6
    tk.close()
    tv = make(copy=False)
    tv.reset()
    first = tv.step(torch.from_numpy(acts[0]).cuda())[0]
    second = tv.step(torch.from_numpy(acts[1]).cuda())[0]
    assert first.data_ptr() == second.data_ptr()
    tv.close()


# --------------------------------------------------------------------------- camera shading (row a14)
def test_render_kernel_shades_like_the_oracle(tmp_path):
    """The ray kernel evaluates the fixed-function lighting equation in single precision, the oracle in double
    (tests/test_render_shading.py pins the oracle to the documented equation): on the spot-light scene, with a material
    and with a directional light, no pixel differs by more than one level, and almost none differs at all; on the
    2-agent level the images carry the light's footprint (the floor is darker far from the light)."""
    from tests.test_render_shading import SCENE
    cases = [dict(light='<light diffuse=".5 .5 .5" pos="0 0 3" dir="0 0 -1"/>', visual="", material=""),
             dict(light='<light diffuse=".5 .5 .5" pos="0 0 3" dir="0 0 -1"/>', visual="", material='material="shiny"'),
             dict(light='<light directional="true" dir="1 0 -1" diffuse=".6 .6 .6"/>',
                  visual='<visual><headlight ambient=".2 .2 .2"/></visual>', material="")]
    for k, case in enumerate(cases):
        path = tmp_path / f"scene{k}.xml"
        path.write_text(SCENE.format(**case))
        packed = blob.pack(mjcf.compile_mjcf(str(path)))
        h = _capi.Handle(packed, 3)
        h.reset()
        ora = OracleEnv(packed)
        for size in (65, 64):
            got = h.render(size, size).astype(int)
            ref = ora.render(0, size, size).reshape(size, size, 3).astype(int)
            assert np.abs(got[1, 0] - ref).max() <= 1, (k, size)
            assert (np.abs(got[1, 0] - ref).max(axis=-1) > 0).mean() < 0.01, (k, size)
            assert k == 2 or len(np.unique(ref.reshape(-1, 3), axis=0)) > 20    # a graded image (the directional case is flat)
        h.close(); ora.close()


def test_render_kernel_casts_the_oracles_shadows(tmp_path):
    """Row a14, shadows: the lights' shadow rays in the ray kernel (fp32, occluders culled once per block and light)
    against the oracle's (fp64, every geom): a sphere over the floor on and off the spot light's axis, a box and a capsule
    between light and floor, a directional light, castshadow off.  Away from the shadows' and the silhouettes' edges
    (where an fp32 ray may fall on the other side) the images agree to one level; the shadow itself is there (the floor
    under the sphere is as dark as the floor outside the light's cone)."""
    from tests.test_render_shading import SHADOW_SCENE
    scenes = [SHADOW_SCENE.format(x=0.0, h=1.5, r=0.3, light=""), SHADOW_SCENE.format(x=1.0, h=1.5, r=0.3, light=""),
              SHADOW_SCENE.format(x=0.0, h=1.5, r=0.3, light='castshadow="false"'),
              SHADOW_SCENE.format(x=0.4, h=1.0, r=0.2, light='directional="true"').replace('dir="0 0 -1"', 'dir="0.3 0.1 -1"'),
              SHADOW_SCENE.format(x=-0.8, h=0.7, r=0.25, light="").replace(
                  "</worldbody>", '<body pos="0.9 0.5 1.2" euler="20 30 0"><freejoint/><geom type="box" size="0.3 0.15 0.1" '
                  'rgba="0 1 0 1"/></body><body pos="0.2 -1.0 0.9" euler="0 70 20"><freejoint/><geom type="capsule" '
                  'size="0.08 0.4" rgba="0 0 1 1"/></body></worldbody>')]
    for k, text in enumerate(scenes):
        path = tmp_path / f"shadow{k}.xml"
        path.write_text(text)
        packed = blob.pack(mjcf.compile_mjcf(str(path)))
        h = _capi.Handle(packed, 2)
        h.reset()
        ora = OracleEnv(packed)
        for size in (129, 64):
            got = h.render(size, size).astype(int)[1, 0]
            ref = ora.render(0, size, size).reshape(size, size, 3).astype(int)
            differ = np.abs(got - ref).max(axis=-1)
            assert (differ > 1).mean() < 0.004, (k, size, (differ > 1).mean())        # edge pixels only
            assert (differ > 0).mean() < 0.015, (k, size)
            if k == 0:
                mid = size // 2
                col = mid + int(round(0.5 / (2 * np.tan(np.radians(22.5)) * 10 / size)))
                assert np.array_equal(got[mid, col], [89, 102, 115]) and np.array_equal(ref[mid, col], [89, 102, 115])
            if k == 2:
                assert (differ > 1).sum() <= 8
        h.close(); ora.close()


def test_the_boxes_tight_culls_change_no_pixel(monkeypatch):
    """The ray kernel drops a box from a block's candidates when it lies wholly behind one of the block's four frustum
    planes, and from a block's shadow candidates when its slab on one of its own axes stays clear of the hull of the
    block's lit points and the light.  Both are conservative: with them switched off (MJRL_RENDER_LOOSE=1: bounding
    spheres only) every image of a stepped batch is the same, byte for byte -- at 64 x 64, at a size that is not a
    multiple of the block, and with a camera looking along a wall."""
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 24, "agentCameras": True})
    env.reset()
    rng = np.random.default_rng(9)
    for t in range(260):
        env.step({a: rng.uniform(-1, 1, (24, 8)) for a in AGENTS})
    # (some copies moved next to a wall, turned)
    qpos = env._handle.get_field("qpos")
    qpos[:6, 0] = [9.2, -9.3, 0.0, 3.0, 9.0, -9.0]; qpos[:6, 1] = [0.0, 0.0, 4.3, -4.4, 4.2, -4.2]
    qpos[:6, 3:7] = [[1, 0, 0, 0], [0.7071, 0, 0, 0.7071], [0.7071, 0, 0, -0.7071], [0.9239, 0, 0, 0.3827], [0, 0, 0, 1], [0.3827, 0, 0, 0.9239]]
    env._handle.set_field("qpos", qpos)
    env._handle.set_scene_cache(False)
    for size in ((64, 64), (72, 40)):
        monkeypatch.delenv("MJRL_RENDER_LOOSE", raising=False)
        tight = env._handle.render(*size)
        monkeypatch.setenv("MJRL_RENDER_LOOSE", "1")
        loose = env._handle.render(*size)
        assert np.array_equal(tight, loose), size
        assert (tight.astype(int).sum(axis=-1) > 0).mean() > 0.3
    env.close()


# --------------------------------------------------------------------------- levels with more bodies / geoms than lanes
def test_arena_with_73_geoms_on_the_device(tmp_path):
    """74 bodies, 73 geoms: the compiler folds the static bodies into the world (53 bodies left), the kernels take geoms
    64..72 in a second pass.  Both kernel builds follow the oracle (contacts, rows, sweeps, state, rangefinder readings),
    the cameras see the late geoms like the oracle's ray caster does, and a folded body still answers by name."""
    from tests.test_big_levels import big_level_text
    path = tmp_path / "big_arena.xml"
    path.write_text(big_level_text())
    model = mjcf.compile_mjcf(str(path))
    assert (model.nbody, model.ngeom) == (53, 73)
    packed = blob.pack(model)
    for specialize in (True, False):
        h = _capi.Handle(packed, 3, specialize=specialize)
        h.set_scene_cache(True)         # (images of the step's own frames, like the oracle's and the reference's)
        h.reset()
        oras = [OracleEnv(packed) for _ in range(3)]
        rng = np.random.default_rng(2)
        for step in range(150):
            ctrl = rng.uniform(-1, 1, (3, model.nu))
            h.set_field("ctrl", ctrl)
            h.step_device(None, 0, 1)
            for e, o in enumerate(oras):
                o.ctrl[:] = ctrl[e]
                o.step()
            if step % 25 == 24:
                stats = h.get_field("solver_stats")
                for e, o in enumerate(oras):
                    assert (stats[e, 0], stats[e, 1], stats[e, 2]) == (o.ncon, o.nefc, o.niter), (specialize, step, e)
        q = h.get_field("qpos")
        for e, o in enumerate(oras):
            assert np.allclose(q[e], o.qpos, rtol=0, atol=1e-9)
            assert np.allclose(h.get_field("sensordata")[e], o.sensordata, atol=1e-7)
        assert max(o.ncon for o in oras) > 0
        got = h.render(64, 64).astype(int)
        for cam in (0, 3):
            ref = oras[1].render(cam, 64, 64).reshape(64, 64, 3).astype(int)
            assert np.abs(got[1, cam] - ref).max() <= 255 and (np.abs(got[1, cam] - ref).max(axis=-1) > 0).mean() < 0.01
        h.close()
    env = MuJoCoRL({"xmlPath": str(path), "agents": ["sender", "receiver", "agent_3", "agent_4"], "numEnvs": 2})
    env.reset()
    rec = env.get_data("pillar_3")
    assert rec["type"] == "body" and np.allclose(rec["position"], [[-8 + 1.3 * 3, 3.5, 0.3]] * 2)
    assert np.allclose(env.distance("pillar_3", "pillar_4"), np.linalg.norm([1.3, 7.0, 0.0]))
    env.close()


def test_autoreset_with_several_frames_per_step_and_runge_kutta(few_build):
    """A reset-without-step has to hold through every launch of a step: two physics frames per step on the 2-agent level
    (two launches) and the Runge-Kutta level (`ant.xml`: four launches per frame, eight per step).  Next-step autoreset
    through the adapter, every copy against its oracle."""
    from mjrl_amd.wrappers import BatchedVectorEnv
    for level, agents, agent, nobs in (("single_agent.xml", ["sender"], "sender", 30), ("ant.xml", ["torso"], "torso", 29)):
        horizon = 4
        env = MuJoCoRL({"xmlPath": levels.level_path(level), "agents": agents, "numEnvs": 3, "maxSteps": horizon, "skipFrames": 2})
        vec = BatchedVectorEnv(env, agent=agent)
        obs, _ = vec.reset()
        assert obs.shape == (3, nobs)
        oras = [OracleEnv(env._blob) for _ in range(3)]
        first = obs.copy()
        idx = env.agents_action_index[agent]
        nsens = env._compiled.nsensordata
        look = lambda o: np.concatenate([o.sensordata[:nsens], o.qpos, o.qvel])
        rng = np.random.default_rng(8)
        t, pending = 0, False
        for step in range(14):
            act = rng.uniform(-1, 1, (3, len(idx)))
            obs, rew, term, trunc, _ = vec.step(act)
            if pending:
                assert np.allclose(obs, first, atol=1e-12) and not trunc.any(), (level, step)
                for o in oras:
                    o.reset()
                pending, t = False, 0
                continue
            for e, o in enumerate(oras):
                o.ctrl[idx] = act[e]
                o.step(); o.step()
            t += 1
            assert np.allclose(obs, np.stack([look(o) for o in oras]), atol=1e-9), (level, step)
            if t == horizon + 1:
                assert trunc.all()
                pending = True
            else:
                assert not trunc.any()
        vec.close()


# --------------------------------------------------------------------------- which frames the cameras draw
def test_render_draws_the_frames_of_the_last_forward_pass(few_build):
    """mjv_updateScene(model, data, ...) (mujoco_parent.py:533) reads the geom / camera / light frames out of MjData as the
    last forward pass left them: after mj_step they are one integration older than qpos.  With the scene cache on
    (agentCameras turns it on) the device images are those of the step's own frames -- the oracle's after the same
    steps, not the oracle's after a fresh forward pass --, a copy reset inside a launch or by the host shows the reset
    image, and a state written by hand is drawn at its own qpos."""
    import torch
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    packed = blob.pack(model)
    h = _capi.Handle(packed, 4)
    h.set_scene_cache(True)
    h.reset()
    oras = [OracleEnv(packed) for _ in range(4)]

    def same(got, ref, frac=0.004):
        got, ref = got.astype(int), np.asarray(ref).reshape(got.shape).astype(int)
        return (np.abs(got - ref).max(axis=-1) > 0).mean() < frac

    reset_images = h.render(64, 64)
    assert all(same(reset_images[e, cam], oras[e].render(cam, 64, 64)) for e in range(4) for cam in range(2))
    rng = np.random.default_rng(21)
    for _ in range(60):
        ctrl = rng.uniform(-1, 1, (4, model.nu))
        h.set_field("ctrl", ctrl)
        h.step_device(None, 0, 1)
        for e, o in enumerate(oras):
            o.ctrl[:] = ctrl[e]
            o.step()
    images = h.render(64, 64)
    for e, o in enumerate(oras):
        assert np.allclose(h.get_field("qpos")[e], o.qpos, rtol=0, atol=1e-9)
        for cam in range(2):
            assert same(images[e, cam], o.render(cam, 64, 64)), (e, cam)
    # ... and NOT the image of the integrated state (the agents are moving: one integration shifts their cameras' views)
    fresh = OracleEnv(packed)
    fresh.qpos[:] = oras[0].qpos
    fresh.forward()
    moved = [(np.abs(fresh.render(cam, 64, 64).reshape(64, 64, 3).astype(int) - images[0, cam].astype(int)).max(axis=-1) > 0).mean()
             for cam in range(2)]
    assert max(moved) > 0.0, moved
    # copy 1 is reset inside the launch without a physics frame (flag 2), copy 2 reset and stepped (flag 1)
    mask = torch.tensor([0, 2, 1, 0], dtype=torch.uint8, device="cuda")
    h.set_step_reset_mask(mask.data_ptr())
    h.step_device(None, 0, 1)
    h.set_step_reset_mask(None)
    oras[1].reset()
    oras[2].reset(); oras[2].ctrl[:] = 0; oras[2].step()
    oras[0].step(); oras[3].step()
    after = h.render(64, 64)
    assert np.array_equal(after[1], reset_images[1])
    for e in (0, 2, 3):
        assert all(same(after[e, cam], oras[e].render(cam, 64, 64)) for cam in range(2)), e
    # a masked reset by the host
    h.reset(np.array([0, 0, 0, 1], np.uint8))
    again = h.render(64, 64)
    assert np.array_equal(again[3], reset_images[3]) and np.array_equal(again[:3], after[:3])
    # a state written by hand is drawn at its own qpos
    q = h.get_field("qpos")
    h.set_field("qpos", q)
    by_hand = h.render(64, 64)
    fresh.qpos[:] = q[0]
    fresh.forward()
    assert all(same(by_hand[0, cam], fresh.render(cam, 64, 64)) for cam in range(2))
    fresh.close(); h.close()
    for o in oras:
        o.close()


# --------------------------------------------------------------------------- the two specialised builds of a model
@pytest.mark.parametrize("few", ["0", "1"])
def test_agent_dropped_onto_agent_in_both_builds(few, monkeypatch):
    """Rows that couple the two agents' trees (one ant dropped onto the other) through the residual-form schedule solver,
    in the build for full batches and in the build for batches of at most one wave per SIMD (mjrl_size "few": schedules
    of 17..32 positions then take the two-position form, 17+ rows in a tree the two-rows-per-lane solver): the oracle's
    contact, row and sweep counts every step, its trajectory at the end."""
    monkeypatch.setenv("MJRL_FEW", few)
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    packed = blob.pack(model)
    h = _capi.Handle(packed, 2)
    assert h.size("few") == int(few)
    free = [j for j in range(model.njnt) if model.jnt_type[j] == 0]
    a0, a1 = (int(model.jnt_qposadr[j]) for j in free)
    q = model.qpos0.copy()
    q[a1:a1 + 3] = q[a0:a0 + 3] + np.array([0.15, 0.1, 0.55])
    h.reset()
    h.set_field("qpos", np.tile(q, (2, 1)))
    oras = [OracleEnv(packed) for _ in range(2)]
    for o in oras:
        o.qpos[:] = q
    rng = np.random.default_rng(8)
    coupled = longest = 0
    for k in range(120):
        ctrl = rng.uniform(-1, 1, (2, model.nu))
        h.set_field("ctrl", ctrl)
        h.step_device(None, 0, 1)
        stats = h.get_field("solver_stats")
        for e, o in enumerate(oras):
            o.ctrl[:] = ctrl[e]
            o.step()
            assert (stats[e, 0], stats[e, 1], stats[e, 2]) == (o.ncon, o.nefc, o.niter), (few, k, e)
        coupled += oras[0].ncon > 0
        longest = max(longest, oras[0].nefc)
    assert coupled > 20 and longest > 20
    assert np.allclose(h.get_field("qpos"), np.stack([o.qpos for o in oras]), rtol=0, atol=1e-9)
    h.close()
    for o in oras:
        o.close()


def test_joint_springs_through_the_c_abi(few_build):
    """Joint stiffness / springref on the GPU (specialised and generic kernel), against the oracle: counts every step,
    states at the end; the springs hold the limbs where -k (q - springref) balances gravity and contact."""
    from tests.test_emu_parity import SPRUNG
    model = mjcf.compile_mjcf_string(SPRUNG)
    packed = blob.pack(model)
    for specialize in (True, False):
        h = _capi.Handle(packed, 3, specialize=specialize)
        h.reset()
        ora = OracleEnv(packed)
        for step in range(700):
            h.step_host(None, 1)
            ora.step()
            stats = h.get_field("solver_stats")
            assert (stats[:, 0] == ora.ncon).all() and (stats[:, 1] == ora.nefc).all() and (stats[:, 2] == ora.niter).all(), step
        assert ora.ncon > 0
        assert np.abs(h.get_field("qpos") - ora.qpos).max() < 1e-9 and np.abs(h.get_field("qvel") - ora.qvel).max() < 1e-8
        h.close(); ora.close()


@pytest.mark.parametrize("seed", [7003, 7020, 9001, 9104])
def test_random_scenes_with_sensors_through_the_c_abi(seed):
    """Random scenes with a site and one to three sensors per body (tests/test_fuzz_scenes.py): the readings of every step
    against the oracle, generic and specialised kernel.  The sensors' constants reach the kernel as lane records built
    on the HOST by the library (not by the emulation's build of the same source): in 7003 and 7020 a touch sensor's
    contact lies outside its site's sphere, so the reading depends on the site's size -- which the library's table held
    as 0 when its writer read the records through an int pointer (tools/parity_fuzz.py with sensors found it)."""
    from tests.test_fuzz_scenes import random_scene
    model = mjcf.compile_mjcf_string(random_scene(np.random.default_rng(seed), sensors=True), nconmax=24, njmax=120)
    packed = blob.pack(model)
    for specialize in (False, True):
        h = _capi.Handle(packed, 2, specialize=specialize)
        h.reset()
        ora = OracleEnv(packed)
        qvel = h.get_field("qvel")
        for j in range(model.njnt):
            if model.jnt_type[j] == mjcf.JNT_FREE:
                qa, da = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
                ora.qvel[da:da + 2] = -2.0 * ora.qpos[qa:qa + 2]
                qvel[:, da:da + 2] = -2.0 * model.qpos0[qa:qa + 2]
        h.set_field("qvel", qvel)
        touched = 0.0
        for step in range(260):
            h.step_host(None, 1)
            ora.step()
            sd = h.get_field("sensordata")
            assert np.allclose(sd, ora.sensordata[None, :], rtol=1e-8, atol=1e-8), (seed, specialize, step, sd[0], ora.sensordata)
            touched = max(touched, float(np.abs(ora.sensordata[np.asarray(model.sensor_adr)[np.asarray(model.sensor_type) == mjcf.SENS_TOUCH]]).max(initial=0.0)))
        if seed in (7003, 7020):
            assert touched > 0
        h.close(); ora.close()


@pytest.mark.parametrize("seed", [3000, 3007, 3010, 3104])
def test_random_articulated_scenes_through_the_c_abi(seed, few_build):
    """Random trees with joint limits, springs, damping, armature and motors (tests/test_fuzz_scenes.py), controls drawn
    beyond their clamp every ten steps: counts of every step and the states at the end against the oracle, generic and
    specialised kernel."""
    from tests.test_fuzz_scenes import random_articulated_scene
    xml, nu = random_articulated_scene(np.random.default_rng(seed))
    model = mjcf.compile_mjcf_string(xml, nconmax=32, njmax=160)
    packed = blob.pack(model)
    for specialize in (False, True):
        h = _capi.Handle(packed, 2, specialize=specialize)
        h.reset()
        ora = OracleEnv(packed)
        crng = np.random.default_rng(seed + 1)
        for step in range(240):
            if step % 10 == 0 and nu:
                ctrl = crng.uniform(-1.3, 1.3, nu)
                ora.ctrl[:nu] = ctrl
                h.set_field("ctrl", np.tile(ctrl, (2, 1)))
            h.step_host(None, 1)
            ora.step()
            stats = h.get_field("solver_stats")
            assert (stats[:, 0] == ora.ncon).all() and (stats[:, 1] == ora.nefc).all() and (stats[:, 2] == ora.niter).all(), (seed, step)
        assert np.abs(h.get_field("qpos") - ora.qpos).max() < 1e-9 and np.abs(h.get_field("qvel") - ora.qvel).max() < 1e-8
        h.close(); ora.close()


def test_ray_kernel_on_random_scenes_against_the_oracle(monkeypatch):
    """Random scenes with a camera on every body and two fixed ones, coloured geoms, one or two lights (spot / directional,
    with and without shadows, one of them riding on a body), after 40 to 280 steps (tools/render_fuzz.py runs 150 of
    them): every image against the oracle's -- apart from one-pixel shifts of edges and steep gradients (fp32 rays
    against fp64 rays) no pixel is off by more than two levels -- and, exactly, against the same kernel with its tight
    culls switched off."""
    from tests.test_fuzz_scenes import random_scene, unexplained
    images = 0
    for seed in range(5000, 5016):
        model = mjcf.compile_mjcf_string(random_scene(np.random.default_rng(seed), cameras=True), nconmax=24, njmax=120)
        packed = blob.pack(model)
        h = _capi.Handle(packed, 2, specialize=False)
        h.set_scene_cache(True)
        h.reset()
        ora = OracleEnv(packed)
        steps = 40 + 60 * (seed % 5)
        for _ in range(steps):
            h.step_host(None, 1)
        ora.step(steps)
        for w, hh in ((64, 64), (72, 40)):
            monkeypatch.delenv("MJRL_RENDER_LOOSE", raising=False)
            got = h.render(w, hh)
            monkeypatch.setenv("MJRL_RENDER_LOOSE", "1")
            assert np.array_equal(got, h.render(w, hh)), (seed, w)
            assert np.array_equal(got[0], got[1])
            for cam in range(model.ncam):
                ref = ora.render(cam, w, hh).reshape(hh, w, 3).astype(int)
                differ = np.abs(got[0, cam].astype(int) - ref).max(axis=-1)
                assert (differ > 1).mean() < 0.02, (seed, w, cam)
                assert unexplained(got[0, cam].astype(int), ref).sum() <= 2, (seed, w, cam)
                images += 1
        h.close(); ora.close()
    assert images > 100


def test_more_sensors_than_lanes_through_the_c_abi(few_build):
    """84 sensors on two free bodies (tests/test_fuzz_scenes.many_sensors_scene): sensors 0..63 come from the lane records the
    library builds on the host, sensors 64.. from the model in the sensor stage's second pass -- every reading of every step
    against the oracle, generic and specialised kernel."""
    from tests.test_fuzz_scenes import many_sensors_scene
    model = mjcf.compile_mjcf_string(many_sensors_scene(), nconmax=24, njmax=120)
    packed = blob.pack(model)
    for specialize in (False, True):
        h = _capi.Handle(packed, 2, specialize=specialize)
        h.reset()
        ora = OracleEnv(packed)
        assert np.allclose(h.get_field("sensordata"), ora.sensordata[None, :], rtol=1e-8, atol=1e-8)
        for step in range(300):
            h.step_host(None, 1)
            ora.step()
            sd = h.get_field("sensordata")
            assert np.allclose(sd, ora.sensordata[None, :], rtol=1e-8, atol=1e-8), (specialize, step)
        assert np.abs(h.get_field("qpos") - ora.qpos).max() < 1e-9
        h.close(); ora.close()


def test_two_motors_on_one_joint_through_the_c_abi(few_build):
    """A dof with two motors (dof_actid -2: the smooth stage scans the actuator list), controls beyond their clamps: states
    against the oracle, generic and specialised kernel."""
    from tests.test_fuzz_scenes import TWO_MOTORS_ONE_JOINT
    model = mjcf.compile_mjcf_string(TWO_MOTORS_ONE_JOINT)
    packed = blob.pack(model)
    for specialize in (False, True):
        h = _capi.Handle(packed, 2, specialize=specialize)
        h.reset()
        ora = OracleEnv(packed)
        rng = np.random.default_rng(3)
        for step in range(200):
            if step % 8 == 0:
                ctrl = rng.uniform(-1.5, 1.5, 3)
                ora.ctrl[:3] = ctrl
                h.set_field("ctrl", np.tile(ctrl, (2, 1)))
            h.step_host(None, 1)
            ora.step()
        assert np.abs(h.get_field("qpos") - ora.qpos).max() < 1e-9 and np.abs(h.get_field("qvel") - ora.qvel).max() < 1e-8
        assert np.abs(ora.qvel[:2]).max() > 0.1          # the arm did move
        h.close(); ora.close()
