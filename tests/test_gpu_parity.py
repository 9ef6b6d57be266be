"""Parity tests proper: the HIP path, called through the C-ABI (libmjrl_hip.so via ctypes), against the CPU
oracle on the same seeded inputs, plus size-independent properties at BASELINE.json's full batch sizes.

Bars (BASELINE.json north_star): body/agent indices, done and truncation flags bit-exact; qpos/qvel within
1e-5 relative over 1000 steps.  Observed agreement is ~1e-12, so the short runs assert 1e-9.
"""
import os

import numpy as np
import pytest

from mjrl_amd import _capi, blob, levels, mjcf
from mjrl_amd.mujoco_rl import MuJoCoRL
from oracle.oracle import OracleEnv

pytestmark = pytest.mark.gpu

AGENTS = ["sender", "receiver"]


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def make(level, n_env, **kw):
    model = mjcf.compile_mjcf(levels.level_path(level), **kw)
    packed = blob.pack(model)
    handle = _capi.Handle(packed, n_env)
    handle.reset()
    return model, packed, handle


def test_reset_state_and_forward_pass():
    model, packed, h = make("two_agent.xml", 5)
    ora = OracleEnv(packed)
    assert np.array_equal(h.get_field("qpos"), np.tile(model.qpos0, (5, 1)))
    assert np.array_equal(h.get_field("qvel"), np.zeros((5, model.nv)))
    assert np.allclose(h.get_field("qacc_warmstart"), ora.qacc_warmstart, atol=1e-10)
    assert np.allclose(h.get_field("sensordata"), ora.sensordata, atol=1e-12)
    assert np.array_equal(h.get_field("timestep"), np.zeros(5, np.int32))


@pytest.mark.parametrize("level,steps", [("two_agent.xml", 400), ("single_agent.xml", 300), ("two_agent_3sensors.xml", 300),
                                         ("four_agent.xml", 300), ("sensor_touch.xml", 150), ("sensor_accelerometer.xml", 150),
                                         ("sensor_rangefinder.xml", 150), ("sensor_framexaxis.xml", 150)])
def test_trajectory_parity_through_the_c_abi(level, steps, few_build):
    n_env = 6
    model, packed, h = make(level, n_env)
    oras = [OracleEnv(packed) for _ in range(n_env)]
    rng = np.random.default_rng(2)
    for _ in range(steps):
        ctrl = rng.uniform(-1, 1, (n_env, max(model.nu, 1)))
        if model.nu:
            h.set_field("ctrl", ctrl[:, :model.nu])
        h.step_host(None, 1)
        for e, o in enumerate(oras):
            o.ctrl[:model.nu] = ctrl[e, :model.nu]
            o.step()
    assert max(o.ncon for o in oras) > 0
    oq, ov = np.stack([o.qpos for o in oras]), np.stack([o.qvel for o in oras])
    assert rel(h.get_field("qpos"), oq) < 1e-9
    assert rel(h.get_field("qvel"), ov) < 1e-9
    assert np.allclose(h.get_field("sensordata"), np.stack([o.sensordata for o in oras]), atol=1e-7)
    assert not h.query("warn").any()                      # no cap (contacts, rows, work items) was hit
    ncon = h.query("ncon")[:, 0]
    assert np.array_equal(ncon, [OracleFwd(o) for o in oras])


def OracleFwd(o):
    o.forward()
    return o.ncon


def test_thousand_step_drift_stays_inside_the_north_star_tolerance(few_build):
    """qpos/qvel within 1e-5 relative over 1000 steps of the 2-agent level with random actions (the drift curve is
    printed; it is the evidence DESIGN.md quotes)."""
    n_env = 4
    model, packed, h = make("two_agent.xml", n_env)
    oras = [OracleEnv(packed) for _ in range(n_env)]
    rng = np.random.default_rng(9)
    worst, curve = 0.0, []
    for step in range(1000):
        ctrl = rng.uniform(-1, 1, (n_env, model.nu))
        h.set_field("ctrl", ctrl)
        h.step_host(None, 1)
        for e, o in enumerate(oras):
            o.ctrl[:] = ctrl[e]
            o.step()
        if step % 100 == 99:
            oq, ov = np.stack([o.qpos for o in oras]), np.stack([o.qvel for o in oras])
            rq, rv = rel(h.get_field("qpos"), oq), rel(h.get_field("qvel"), ov)
            worst = max(worst, rq, rv)
            curve.append(f"step {step + 1}: qpos rel {rq:.2e}  qvel rel {rv:.2e}")
    print("\n".join(curve))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "drift_curve.txt"), "w") as fh:
            fh.write("\n".join(curve) + "\n")
    assert worst < 1e-5


def test_scatter_gather_flags_through_the_env_class():
    """Indices and flags are bit-exact: ctrl lands where the reference's table says, observations are
    sensordata|qpos|qvel, truncation fires on call max_steps+1 (mujoco_rl.py:279,288)."""
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 3, "maxSteps": 4})
    assert env.agents_action_index == {"sender": [2, 3, 4, 5, 6, 7, 0, 1], "receiver": [10, 11, 12, 13, 14, 15, 8, 9]}
    obs, infos = env.reset()
    assert obs["sender"].shape == (3, 59) and infos == {"sender": {}, "receiver": {}}
    oras = [OracleEnv(env._blob) for _ in range(3)]
    rng = np.random.default_rng(4)
    for call in range(6):
        action = {a: rng.uniform(-1, 1, (3, 8)) for a in AGENTS}
        obs, rew, term, trunc, info = env.step(action)
        for e, o in enumerate(oras):
            for a in AGENTS:
                o.ctrl[env.agents_action_index[a]] = action[a][e]
            o.step()
        assert np.array_equal(env._handle.get_field("ctrl"), np.stack([o.ctrl for o in oras]))
        for k, a in enumerate(AGENTS):
            expect = np.stack([np.concatenate([o.sensordata[[k]], o.qpos, o.qvel]) for o in oras])
            assert np.allclose(obs[a], expect, atol=1e-10)
            assert np.array_equal(rew[a], np.zeros(3)) and not term[a].any()
            assert trunc[a].all() == (call >= 4)
        assert "__all__" in trunc and "__all__" not in term
    assert np.array_equal(env._handle.get_field("timestep"), np.full(3, 6, np.int32))
    env.close()


def test_single_copy_has_the_reference_shapes_and_runs_host_plugins(few_build):
    calls = []

    class Language:
        """The README's language channel (README.md:109-136) in its 4-tuple form."""
        def __init__(self, env):
            self.env = env
            self.observation_space = {"low": [0], "high": [3]}
            self.action_space = {"low": [0], "high": [3]}

        def dynamic(self, agent, actions):
            calls.append(agent)
            store = self.env.data_store
            store[agent]["utterance"] = int(actions[0])
            other = [a for a in self.env.agents if a != agent][0]
            heard = store[other].get("utterance", 0)
            return 0, np.array([heard]), False, {}

    def reward(env, agent):
        return float(env.data.qpos[2] if agent == "sender" else env.data.qpos[17])

    def done(env, agent):
        return bool(env.data.qpos[2] < 0.2)

    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "environmentDynamics": [Language],
                    "rewardFunctions": [reward], "doneFunctions": [done]})
    assert env.action_space("sender").shape == (9,) and env.observation_space("sender").shape == (60,)
    assert env.action_routing == {"physical": [0, 8], "dynamic": {"Language": [8, 9]}}
    obs, infos = env.reset()
    assert obs["sender"].shape == (60,) and env.data_store == {"sender": {}, "receiver": {}}
    calls.clear()
    act = {"sender": np.r_[np.zeros(8), 2.0], "receiver": np.r_[np.zeros(8), 1.0]}
    obs, rew, term, trunc, info = env.step(act)
    assert calls == ["sender", "receiver"]
    assert obs["sender"][-1] == 0 and obs["receiver"][-1] == 2     # the receiver already hears this step's utterance
    obs, rew, term, trunc, info = env.step(act)
    assert obs["sender"][-1] == 1
    assert rew["sender"] == pytest.approx(obs["sender"][1 + 2]) and isinstance(term["sender"], bool)
    assert term["__all__"] is False and info["sender"] == {"Language": {}}
    assert env.distance("sender", "receiver") == pytest.approx(np.linalg.norm(env.data.body("sender").xipos - env.data.body("receiver").xipos))
    assert env.collision("sender_geom", "receiver_geom") is False
    assert env.get_data("sender")["type"] == "body" and env.get_data("border1_geom")["type"] == "geom"
    env.close()


def test_full_batch_properties():
    """4096 copies (BASELINE.json metric size): finite, deterministic run to run, independent of batch position,
    identical copies stay identical, and a masked reset touches only the masked copies."""
    n_env = 4096
    model, packed, h = make("two_agent.xml", n_env)
    rng = np.random.default_rng(1)
    base = rng.uniform(-1, 1, (200, 16, model.nu))          # 16 distinct action streams, tiled over the batch
    def run(handle, perm):
        handle.reset()
        for t in range(200):
            ctrl = np.tile(base[t], (n_env // 16, 1))[perm]
            handle.set_field("ctrl", ctrl)
            handle.step_device(None, 0, 1)
        return handle.get_field("qpos"), handle.get_field("qvel")
    ident = np.arange(n_env)
    q1, v1 = run(h, ident)
    assert np.isfinite(q1).all() and np.isfinite(v1).all()
    assert np.array_equal(q1[:16], q1[16:32]) and np.array_equal(q1[:16], q1[-16:])      # same stream -> same bits
    q2, v2 = run(h, ident)
    assert np.array_equal(q1, q2) and np.array_equal(v1, v2)                              # run-to-run determinism
    perm = rng.permutation(n_env)
    q3, _ = run(h, perm)
    assert np.array_equal(q3, q1[perm])                                                   # position independence
    oras = [OracleEnv(packed) for _ in range(4)]
    for t in range(200):
        for e, o in enumerate(oras):
            o.ctrl[:] = base[t, e]
            o.step()
    assert rel(q1[:4], np.stack([o.qpos for o in oras])) < 1e-9
    mask = np.zeros(n_env, np.uint8)
    mask[::7] = 1
    h.reset(mask)
    q4 = h.get_field("qpos")
    assert np.array_equal(q4[::7], np.tile(model.qpos0, (len(q4[::7]), 1)))
    keep = np.ones(n_env, bool)
    keep[::7] = False
    assert np.array_equal(q4[keep], q3[keep])      # the last run on this handle was the permuted one


def test_config_two_batch_runs_the_few_copies_build_and_follows_the_oracle(monkeypatch):
    """1024 copies of the 2-agent level (BASELINE config 2; a quarter of the chip's wave slots): left to itself the
    library attaches the build for batches of at most one wave per SIMD.  Finite, deterministic, identical action
    streams give identical bits wherever they sit in the batch, eight copies against their oracles with the same
    per-copy contact / row / sweep counts."""
    monkeypatch.delenv("MJRL_FEW", raising=False)
    n_env, steps = 1024, 300
    model, packed, h = make("two_agent.xml", n_env)
    assert h.size("few") == 1 and h.kernel == "specialised"
    rng = np.random.default_rng(11)
    base = rng.uniform(-1, 1, (steps, 8, model.nu))
    oras = [OracleEnv(packed) for _ in range(8)]
    def run():
        h.reset()
        counts = []
        for t in range(steps):
            h.set_field("ctrl", np.tile(base[t], (n_env // 8, 1)))
            h.step_device(None, 0, 1)
            counts.append(h.get_field("solver_stats")[:8, :3].copy())
        return h.get_field("qpos"), h.get_field("qvel"), counts
    q1, v1, counts = run()
    assert np.isfinite(q1).all() and np.isfinite(v1).all()
    assert np.array_equal(q1[:8], q1[8:16]) and np.array_equal(q1[:8], q1[-8:])
    q2, v2, _ = run()
    assert np.array_equal(q1, q2) and np.array_equal(v1, v2)
    for t in range(steps):
        for e, o in enumerate(oras):
            o.ctrl[:] = base[t, e]
            o.step()
            assert tuple(counts[t][e]) == (o.ncon, o.nefc, o.niter), (t, e)
    assert max(o.ncon for o in oras) > 0
    assert rel(q1[:8], np.stack([o.qpos for o in oras])) < 1e-9
    assert rel(v1[:8], np.stack([o.qvel for o in oras])) < 1e-9
    assert h.cap_overflows() == (0, 0)


def test_step_batched_with_torch_tensors_stays_on_device():
    import torch
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 64})
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    act = torch.rand((64, 2, 8), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    obs, rew, term, trunc = env.step_batched(act)
    torch.cuda.synchronize()
    ora = OracleEnv(env._blob)
    a0 = act[0].cpu().numpy()
    for k, a in enumerate(AGENTS):
        ora.ctrl[env.agents_action_index[a]] = a0[k]
    ora.step()
    expect = np.concatenate([ora.sensordata[[0]], ora.qpos, ora.qvel])
    assert np.allclose(obs[0, 0].cpu().numpy(), expect, atol=1e-10)
    assert obs.shape == (64, 2, 59) and not term.any().item() and not trunc.any().item()
    env.close()


def test_fused_vocabulary_equals_the_host_plugin_loop(few_build):
    """Config 3: Language channel + rangefinder obs, with a target-distance reward and done.  The same plugins run
    once as ops of the step kernel and once through the host plugin loop (fusedPlugins False); both sit on the
    same GPU physics, so every output must agree (utterances exactly)."""
    from mjrl_amd.dynamics import Language, TargetDistanceReward, TargetReached

    def make_env(fused):
        return MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 5, "maxSteps": 12,
                         "environmentDynamics": [Language], "fusedPlugins": fused,
                         "rewardFunctions": [TargetDistanceReward("reference", mode="delta", scale=2.0)],
                         "doneFunctions": [TargetReached("reference", 3.4)]})
    fused, host = make_env(True), make_env(False)
    assert fused._program is not None and host._program is None
    assert fused.observation_space("sender").shape == (60,) and fused.action_space("sender").shape == (9,)
    np.random.seed(0)
    o1, _ = fused.reset()
    np.random.seed(0)
    o2, _ = host.reset()
    assert o1["sender"].shape == o2["sender"].shape == (5, 60)
    rng = np.random.default_rng(3)
    for step in range(16):
        action = {a: np.concatenate([rng.uniform(-1, 1, (5, 8)), rng.uniform(0, 3, (5, 1))], axis=1) for a in AGENTS}
        f, h = fused.step(action), host.step(action)
        for a in AGENTS:
            assert np.array_equal(f[0][a][:, 59], h[0][a][:, 59])              # what was heard
            assert np.allclose(f[0][a], h[0][a], atol=1e-12)
            assert np.allclose(f[1][a], h[1][a], atol=1e-10)
            assert np.array_equal(f[2][a], h[2][a]) and np.array_equal(f[3][a], h[3][a])
        assert np.array_equal(f[2]["__all__"], h[2]["__all__"]) and f[4] == h[4]
    assert f[3]["sender"].all() and np.abs(f[1]["sender"]).max() > 0 and f[2]["sender"].any() != f[2]["receiver"].any()
    assert np.array_equal(fused.device_store["sender"]["utterance"], np.trunc(action["sender"][:, 8]))
    fused.reset()
    assert np.isnan(fused.device_store["sender"]["utterance"]).all()
    fused.close()
    host.close()


def test_agent_cameras_match_the_oracle_ray_caster(few_build):
    """Config 5: body-mounted cameras, 64x64x3 uint8 (the input contract of vision/autoencoder.py:13).  Pixel parity
    with the reference's OpenGL output is unpinned; the device ray caster is checked against the
    oracle's on the same states, allowing isolated one-level differences on silhouette edges."""
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 3, "agentCameras": True})
    env.reset()
    oras = [OracleEnv(env._blob) for _ in range(3)]
    rng = np.random.default_rng(5)
    for _ in range(40):
        action = {a: rng.uniform(-1, 1, (3, 8)) for a in AGENTS}
        env.step(action)
        for e, o in enumerate(oras):
            for a in AGENTS:
                o.ctrl[env.agents_action_index[a]] = action[a][e]
            o.step()
    images = env.get_camera_data("sender")
    assert images.shape == (3, 1, 64, 64, 3) and images.dtype == np.uint8
    assert env.get_camera_data("receiver_camera").shape == (3, 64, 64, 3)
    for e, o in enumerate(oras):
        # (both sides draw the frames their last step's forward pass left -- what mjv_updateScene reads out of MjData,
        # mujoco_parent.py:533 --, not kinematics of the integrated qpos: agentCameras turns the scene cache on)
        for cam, agent in enumerate(AGENTS):
            ref = o.render(cam, 64, 64).astype(int)
            got = env.get_camera_data(agent)[e, 0].astype(int)
            differ = np.abs(ref - got).max(axis=-1) > 0
            assert differ.mean() < 0.002 and np.abs(ref - got).max() <= 255
            assert (got.sum(axis=-1) > 0).mean() > 0.2        # the image is not empty
    one = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "agentCameras": True})
    one.reset()
    assert one.get_camera_data("sender").shape == (1, 64, 64, 3)     # the reference's (ncam, W, H, 3)
    env.close()
    one.close()


def test_wrappers():
    from mjrl_amd.wrappers import BatchedVectorEnv, GymnasiumWrapper
    single = MuJoCoRL({"xmlPath": levels.level_path("single_agent.xml"), "agents": ["sender"], "maxSteps": 5})
    gym_env = GymnasiumWrapper(single, "sender")
    obs, infos = gym_env.reset()
    assert obs.shape == gym_env.observation_space.shape == (1 + 15 + 14,)
    out = gym_env.step(gym_env.action_space.sample())
    assert out[0].shape == obs.shape and out[1] == 0 and out[2] is False and out[3] is False and out[4] == {}
    with pytest.raises(Exception, match="too many agents"):
        GymnasiumWrapper(MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS}), "sender")
    single.close()
    vec = BatchedVectorEnv(MuJoCoRL({"xmlPath": levels.level_path("single_agent.xml"), "agents": ["sender"],
                                     "numEnvs": 7, "maxSteps": 5}), autoreset="same_step")      # (the SB3 convention)
    obs, _ = vec.reset()
    assert obs.shape == (7, 30)
    first = obs.copy()
    for step in range(1, 8):
        obs, rew, term, trunc, info = vec.step(np.zeros((7, 8)))
        assert obs.shape == (7, 30) and rew.shape == (7,) and not term.any()
        if step == 6:                      # the horizon: every copy truncated, reset, fresh observation returned
            assert trunc.all() and np.allclose(obs, first) and "final_observation" in info
        else:
            assert not trunc.any()
    vec.close()


def test_errors_are_reported_not_swallowed():
    model, packed, h = make("two_agent.xml", 2)
    with pytest.raises(Exception, match="unknown field"):
        h._check(h._lib.mjrl_get_field(h._h, b"nope", None, 0))
    with pytest.raises(Exception, match="out of range"):
        h.set_scatter_tables([[99]], 0)
    with pytest.raises(Exception, match="blob rejected"):
        _capi.Handle(b"\0" * 400, 1)
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS})
    env.reset()
    with pytest.raises(Exception, match="number of actions"):
        env.step({"sender": np.zeros(3), "receiver": np.zeros(8)})
    env.close()


def test_smoke_entry():
    import __graft_entry__ as entry
    entry.smoke()


def test_specialised_kernel_is_bit_identical_to_the_generic_one():
    """The per-model-shape build of the step kernel (kernel_cache / mjrl_load_kernel) is the same arithmetic."""
    from mjrl_amd import kernel_cache
    n_env = 16
    model = mjcf.compile_mjcf(levels.level_path("two_agent_3sensors.xml"))
    packed = blob.pack(model)
    spec, gen = _capi.Handle(packed, n_env, specialize=True), _capi.Handle(packed, n_env, specialize=False)
    assert spec.kernel == "specialised" and gen.kernel == "generic"
    spec.reset(); gen.reset()
    rng = np.random.default_rng(11)
    for t in range(240):
        ctrl = rng.uniform(-1, 1, (n_env, model.nu))
        frames = 3 if t % 40 == 7 else 1           # a multi-frame step is several launches of the one-frame kernel
        for h in (spec, gen):
            h.set_field("ctrl", ctrl)
            h.step_host(None, frames)
    # (a query without the frame cache is a forward pass, which refreshes the warm start: ask both batches)
    assert spec.query("ncon").max() > 0 and gen.query("ncon").max() > 0
    for field in ("qpos", "qvel", "qacc_warmstart", "sensordata", "timestep"):
        assert np.array_equal(spec.get_field(field), gen.get_field(field)), field
    # a code object built for another shape is refused and the generic kernel stays in place
    other = kernel_cache.code_object(blob.pack(mjcf.compile_mjcf(levels.level_path("single_agent.xml"))))
    with pytest.raises(Exception, match="different model shape"):
        gen.load_kernel(other)
    assert gen.kernel == "generic"
    gen.step_host(None, 1)


def test_general_paths_and_two_chain_rows_on_the_gpu():
    """The paths random play rarely reaches, through the C-ABI: a model without the tree-row lane map (compact rows,
    level-parallel LDS solves, serial sweeps) and hand-posed states whose contacts join two moving bodies -- agent on
    agent (two kinematic trees) and leg on leg inside one agent (ancestor chains that share the torso dofs)."""
    # no lane map
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"), lane_map=False)
    packed = blob.pack(model)
    n_env = 4
    h = _capi.Handle(packed, n_env)
    h.reset()
    oras = [OracleEnv(packed) for _ in range(n_env)]
    rng = np.random.default_rng(8)
    for _ in range(260):
        ctrl = rng.uniform(-1, 1, (n_env, model.nu))
        h.set_field("ctrl", ctrl)
        h.step_host(None, 1)
        for e, o in enumerate(oras):
            o.ctrl[:] = ctrl[e]
            o.step()
    assert max(o.ncon for o in oras) > 0 and not h.query("warn").any()
    assert rel(h.get_field("qpos"), np.stack([o.qpos for o in oras])) < 1e-9
    assert rel(h.get_field("qvel"), np.stack([o.qvel for o in oras])) < 1e-8
    # posed states with two-chain rows
    model, packed, h = make("two_agent.xml", 2)
    tree, gbody = model.body_treeid, model.geom_bodyid
    free = [j for j in range(model.njnt) if model.jnt_type[j] == 0]
    a0, a1 = (int(model.jnt_qposadr[j]) for j in free)
    hinge_q = [int(model.jnt_qposadr[j]) for j in range(model.njnt) if model.jnt_type[j] == 3]
    stacked = model.qpos0.copy()
    stacked[a1:a1 + 3] = stacked[a0:a0 + 3] + np.array([0.15, 0.1, 0.55])      # agent 1 dropped onto agent 0
    probe, folded = OracleEnv(packed), None
    rng = np.random.default_rng(3)
    for _ in range(200):                                                       # legs folded into each other, in the air
        q = model.qpos0.copy()
        q[hinge_q] = rng.uniform(-2.5, 2.5, len(hinge_q))
        q[a0 + 2] += 1.0; q[a1 + 2] += 1.0
        probe.qpos[:] = q; probe.qvel[:] = 0
        probe.forward()
        if any(tree[gbody[c["geom1"]]] == tree[gbody[c["geom2"]]] >= 0 for c in probe.contacts()):
            folded = q
            break
    assert folded is not None
    start = np.stack([stacked, folded])
    h.set_field("qpos", start)
    oras = [OracleEnv(packed) for _ in range(2)]
    seen = [0, 0]
    for e, o in enumerate(oras):
        o.qpos[:] = start[e]
    for _ in range(60):
        h.step_host(None, 1)
        for e, o in enumerate(oras):
            o.step()
            for c in o.contacts():
                t1, t2 = int(tree[gbody[c["geom1"]]]), int(tree[gbody[c["geom2"]]])
                seen[e] += (t1 >= 0 and t2 >= 0 and ((t1 != t2) if e == 0 else (t1 == t2)))
    assert seen[0] > 0 and seen[1] > 0
    assert rel(h.get_field("qpos"), np.stack([o.qpos for o in oras])) < 1e-9
    assert rel(h.get_field("qvel"), np.stack([o.qvel for o in oras])) < 1e-8


@pytest.mark.parametrize("few", ["0", "1"])
def test_wide_register_solver_on_the_gpu(few, monkeypatch):
    """17..32 constraint rows in one kinematic tree (an ant on its four feet with joints at their limits) take the
    32-rows-per-tree register solver (pgs_wide_registers) in the build for full batches and the two-rows-per-lane one
    (pgs_tall_registers) in the build for batches of at most one wave per SIMD (mjrl_size "few"; MJRL_FEW picks the kind
    whatever the batch size)."""
    from tests.test_emu_parity import _pose_with_many_rows_in_one_tree
    monkeypatch.setenv("MJRL_FEW", few)
    model, packed, h = make("two_agent.xml", 3)
    assert h.size("few") == int(few)
    q = _pose_with_many_rows_in_one_tree(model, packed)
    assert q is not None
    start = np.tile(q, (3, 1))
    h.set_field("qpos", start)
    oras = [OracleEnv(packed) for _ in range(3)]
    for o in oras:
        o.qpos[:] = q
    ioff, info_at = h.lds_offset("ints"), h.lds_offset("i_rowinfo")
    rng = np.random.default_rng(4)
    wide_steps = 0
    for k in range(60):
        ctrl = rng.uniform(-1, 1, (3, model.nu))
        h.set_field("ctrl", ctrl)
        img = h.step_debug(None, 0, 1, 0)
        ints = img[:, ioff:ioff + (info_at + model.njmax + 1) // 2 + 1].copy().view(np.int32)
        for e, o in enumerate(oras):
            o.ctrl[:] = ctrl[e]
            o.step()
            assert (ints[e, 1], ints[e, 0], ints[e, 3]) == (o.nefc, o.ncon, o.niter), (k, e)
        trees = (ints[0, info_at:info_at + ints[0, 1]] >> 19) - 2
        per_tree = [int((trees == t).sum()) for t in range(model.ntree)]
        wide_steps += bool((trees >= 0).all() and 16 < max(per_tree) <= 32)
    assert wide_steps > 0
    assert rel(h.get_field("qpos"), np.stack([o.qpos for o in oras])) < 1e-9
    assert rel(h.get_field("qvel"), np.stack([o.qvel for o in oras])) < 1e-8


def test_cap_overflows_are_counted_and_match_the_oracle():
    """A copy that runs into nconmax drops the later contacts in detection order, as the oracle does, and every such
    frame is counted (mjrl_cap_overflows = MuJoCo's mjWARN_CONTACTFULL count); with room for the contacts the count
    stays zero."""
    model, packed, h = make("sensor_touch.xml", 5, nconmax=2, njmax=8)
    assert h.cap_overflows() == (0, 0)
    ora = OracleEnv(packed)
    cut = 0
    for _ in range(40):
        h.step_device(None, 0, 1)
        ora.step()
        cut += ora.warnings & 1
    assert cut > 0
    assert h.cap_overflows() == (5 * cut, 0)
    assert np.allclose(h.get_field("qpos"), np.tile(ora.qpos, (5, 1)), atol=1e-10)
    h.query("warn")                                      # forward-only launches do not count
    assert h.cap_overflows(clear=True) == (5 * cut, 0)
    assert h.cap_overflows() == (0, 0)
    h.close()
    model, packed, h = make("sensor_touch.xml", 5)
    for _ in range(40):
        h.step_device(None, 0, 1)
    assert h.cap_overflows() == (0, 0)
    h.close()


def test_env_class_warns_about_dropped_contacts_at_reset():
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 4, "nconmax": 1, "njmax": 20})
    env.reset()
    acts = np.zeros((4, 2, 8))
    for _ in range(320):                                     # the ants land on four feet each: more than one contact
        env.step_batched(acts)
    with pytest.warns(RuntimeWarning, match="nconmax=1"):
        env.reset()
    assert env.cap_overflows() == (0, 0)
    env.close()


def test_render_block_tiling_at_odd_sizes_and_batch_sizes():
    """The render kernel works in 8x8 pixel blocks, several workgroups per camera, with a block-level cone cull in
    front of the per-ray tests: image sizes that are no multiple of 8 and batch sizes that change the number of
    workgroups per camera must give the same pixels as the oracle's plain per-pixel loop."""
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    packed = blob.pack(model)
    ora = OracleEnv(packed)
    rng = np.random.default_rng(9)
    for _ in range(300):
        ora.ctrl[:] = rng.uniform(-1, 1, model.nu)
        ora.step()
    ora.forward()             # (a state written by hand is drawn at its own qpos on the device: fresh frames here too)
    for n_env, (w, h) in ((1, (37, 21)), (5, (64, 64)), (40, (20, 12))):
        handle = _capi.Handle(packed, n_env)
        handle.reset()
        handle.set_field("qpos", np.tile(ora.qpos, (n_env, 1)))
        got = handle.render(w, h).astype(int)                      # [n_env, ncam, h, w, 3]
        for cam in range(got.shape[1]):
            ref = ora.render(cam, w, h).reshape(h, w, 3).astype(int)
            for e in (0, n_env - 1):
                differ = np.abs(ref - got[e, cam]).max(axis=-1) > 0
                assert differ.mean() < 0.004, (n_env, w, h, cam, differ.mean())
            assert (got[:, cam] == got[0, cam]).all()
        handle.close()


def test_four_agent_schedule_solver_on_the_gpu_and_in_both_builds():
    """Four agents stacked in pairs (rows that couple trees (0,1), (2,3), (1,2) in one step: the aligned schedule with
    two positions per lane in the specialised build, the LDS-resident loop in the generic one) and, in another copy,
    one agent pressed onto the floor (17+ rows in a tree).  Both builds must give the oracle's sweep counts and each
    other's bits."""
    model = mjcf.compile_mjcf(levels.level_path("four_agent.xml"))
    packed = blob.pack(model)
    free = [j for j in range(model.njnt) if model.jnt_type[j] == 0]
    adr = [int(model.jnt_qposadr[j]) for j in free]
    stacked = model.qpos0.copy()
    base = stacked[adr[0]:adr[0] + 3].copy()
    stacked[adr[1]:adr[1] + 3] = base + np.array([0.15, 0.10, 0.55])
    stacked[adr[2]:adr[2] + 3] = base + np.array([1.05, 0.15, 0.00])
    stacked[adr[3]:adr[3] + 3] = base + np.array([1.20, 0.25, 0.55])
    pressed = model.qpos0.copy()
    pressed[adr[0] + 2] -= 0.80
    hinge0 = [int(model.jnt_qposadr[j]) for j in range(model.njnt)
              if model.jnt_type[j] == 3 and model.body_treeid[model.jnt_bodyid[j]] == 0]
    pressed[hinge0] += np.random.default_rng(6).uniform(-0.3, 0.3, len(hinge0))
    start = np.stack([stacked, pressed, model.qpos0])
    handles = [_capi.Handle(packed, 3, specialize=True), _capi.Handle(packed, 3, specialize=False)]
    assert handles[0].kernel == "specialised" and handles[1].kernel == "generic"
    oras = [OracleEnv(packed) for _ in range(3)]
    for h in handles:
        h.reset()
        h.set_field("qpos", start)
    for e, o in enumerate(oras):
        o.qpos[:] = start[e]
    ioff, info_at = handles[0].lds_offset("ints"), handles[0].lds_offset("i_rowinfo")
    rng = np.random.default_rng(7)
    coupled_steps = many_row_steps = 0
    for k in range(45):
        ctrl = rng.uniform(-1, 1, (3, model.nu))
        imgs = []
        for h in handles:
            h.set_field("ctrl", ctrl)
            imgs.append(h.step_debug(None, 0, 1, 0))
        ints = imgs[0][:, ioff:ioff + (info_at + model.njmax + 1) // 2 + 1].copy().view(np.int32)
        for e, o in enumerate(oras):
            o.ctrl[:] = ctrl[e]
            o.step()
            assert (ints[e, 1], ints[e, 0], ints[e, 3]) == (o.nefc, o.ncon, o.niter), (k, e)
            trees = (ints[e, info_at:info_at + ints[e, 1]] >> 19) - 2
            coupled_steps += bool((trees == -2).any())
            many_row_steps += bool((trees >= 0).all() and max((trees == t).sum() for t in range(model.ntree)) > 16)
    assert coupled_steps > 0 and many_row_steps > 0
    for name in ("qpos", "qvel", "qacc_warmstart"):
        assert np.array_equal(handles[0].get_field(name), handles[1].get_field(name)), name
    assert rel(handles[0].get_field("qpos"), np.stack([o.qpos for o in oras])) < 1e-9
    assert rel(handles[0].get_field("qvel"), np.stack([o.qvel for o in oras])) < 1e-8
    for h in handles:
        h.close()
