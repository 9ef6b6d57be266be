"""Round-2 parity tests on the MI355X, through the C-ABI, against the CPU oracle: the freeJoint (qvel-overwrite)
scatter of the reference's own benchmark configuration, the plugin query helpers against the oracle's forward-pass
arrays, the reset image / in-launch reset, full-batch (4096-copy) properties of the 4-agent arena and of the fused
config-3 program, 512-copy camera properties, the per-copy solver statistics, the multi-level re-initialisation hook and
the device-buffer validation of ``step_batched``."""
import random

import numpy as np
import pytest

from mjrl_amd import _capi, blob, levels, mjcf
from mjrl_amd.mujoco_rl import MuJoCoRL
from oracle.oracle import OracleEnv

pytestmark = pytest.mark.gpu

AGENTS = ["sender", "receiver"]
FOUR = ["sender", "receiver", "agent_3", "agent_4"]


def rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def make(level, n_env, **kw):
    model = mjcf.compile_mjcf(levels.level_path(level), **kw)
    packed = blob.pack(model)
    handle = _capi.Handle(packed, n_env)
    handle.reset()
    return model, packed, handle


# --------------------------------------------------------------------------- (i) freeJoint scatter
@pytest.mark.parametrize("level,agents,steps", [("two_agent.xml", AGENTS, 300), ("sensor_touch.xml", ["receiver"], 200)])
def test_free_joint_mode_against_the_oracle(level, agents, steps, few_build):
    """freeJoint=True -- the configuration of the reference's fps_benchmark.py:20 and Testing/sensor_test.py:20: the
    action overwrites qvel[dof, dof+1, dof+5] of the agent's free joint before the physics (mujoco_parent.py:325)."""
    n_env = 5
    env = MuJoCoRL({"xmlPath": levels.level_path(level), "agents": agents, "numEnvs": n_env, "freeJoint": True})
    if level == "two_agent.xml":
        assert env.agents_action_index == {"sender": [0, 1, 5], "receiver": [14, 15, 19]}
    assert env.action_space(agents[0]).shape == (3,)
    env.reset()
    oras = [OracleEnv(env._blob) for _ in range(n_env)]
    rng = np.random.default_rng(11)
    for _ in range(steps):
        action = {a: rng.uniform(-1, 1, (n_env, 3)) for a in agents}
        obs, rew, term, trunc, info = env.step(action)
        for e, o in enumerate(oras):
            for a in agents:
                o.qvel[env.agents_action_index[a]] = action[a][e]
            o.step()
    assert max(o.ncon for o in oras) > 0
    assert rel(env._handle.get_field("qpos"), np.stack([o.qpos for o in oras])) < 1e-9
    assert rel(env._handle.get_field("qvel"), np.stack([o.qvel for o in oras])) < 1e-9
    for k, a in enumerate(agents):
        n_s = len(env.agents_observation_index[a]["sensors"])
        expect = np.stack([np.concatenate([o.sensordata[env.agents_observation_index[a]["sensors"]], o.qpos, o.qvel]) for o in oras])
        assert obs[a].shape == (n_env, n_s + env.model.nq + env.model.nv)
        assert np.allclose(obs[a], expect, atol=1e-9)
    env.close()


# --------------------------------------------------------------------------- (ii) query helpers
def test_queries_return_the_forward_pass_frames_of_the_oracle():
    """get_data / distance / collision read xipos, geom_xpos, geom_xmat and the contact list as MjData holds them after
    mj_step: the frames of the forward pass INSIDE the step, one integration older than qpos (mujoco_parent.py:404-416,
    472-475).  The oracle's arrays after ora_step are exactly those."""
    n_env = 4
    model, packed, h = make("two_agent.xml", n_env)
    h.set_query_cache(True)
    oras = [OracleEnv(packed) for _ in range(n_env)]
    rng = np.random.default_rng(21)
    for step in range(260):
        ctrl = rng.uniform(-1, 1, (n_env, model.nu))
        h.set_field("ctrl", ctrl)
        h.step_host(None, 1)
        for e, o in enumerate(oras):
            o.ctrl[:] = ctrl[e]
            o.step()
        if step in (0, 50, 200, 230, 259):
            for name, field in (("xpos", "xpos"), ("xquat", "xquat"), ("xipos", "xipos"), ("geom_xpos", "geom_xpos")):
                got = h.query(name)
                ref = np.stack([getattr(o, field) for o in oras])
                assert np.allclose(got, ref, atol=1e-10), (name, step)
            gm = h.query("geom_xmat")
            assert np.allclose(gm, np.stack([o.geom_xmat for o in oras]), atol=1e-10)
            ncon = h.query("ncon")[:, 0].astype(int)
            assert np.array_equal(ncon, [o.ncon for o in oras])
            pairs = h.query("contact_geom").astype(int)
            for e, o in enumerate(oras):
                ref = np.array([[c["geom1"], c["geom2"]] for c in o.contacts()]).reshape(-1, 2)
                assert np.array_equal(pairs[e, :o.ncon], ref)
                assert (pairs[e, o.ncon:] == -1).all()
            # the frames are the step's forward pass, not the post-integration state: the oracle's kinematics at the
            # new qpos differ from them
            stats = h.get_field("solver_stats")
            assert np.array_equal(stats[:, 0], [o.ncon for o in oras])
            assert np.array_equal(stats[:, 1], [o.nefc for o in oras])
            assert np.array_equal(stats[:, 2], [o.niter for o in oras])
    assert max(o.ncon for o in oras) > 0
    # a query must not disturb the trajectory (the forward pass it may trigger leaves the warm start alone)
    warm = h.get_field("qacc_warmstart")
    h.set_field("qpos", h.get_field("qpos"))          # invalidates the cache: the next query runs a forward pass
    h.query("xipos")
    assert np.array_equal(h.get_field("qacc_warmstart"), warm)


def test_env_class_queries_match_the_oracle():
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 3,
                    "rewardFunctions": [lambda e, a: 0.0]})        # a host plugin: the query cache is on
    env.reset()
    oras = [OracleEnv(env._blob) for _ in range(3)]
    rng = np.random.default_rng(8)
    for _ in range(230):
        action = {a: rng.uniform(-1, 1, (3, 8)) for a in AGENTS}
        env.step(action)
        for e, o in enumerate(oras):
            for a in AGENTS:
                o.ctrl[env.agents_action_index[a]] = action[a][e]
            o.step()
    names = env._compiled.names
    sb, rb = names["body"].index("sender"), names["body"].index("receiver")
    ref = np.array([np.linalg.norm(o.xipos[sb] - o.xipos[rb]) for o in oras])
    assert np.allclose(env.distance("sender", "receiver"), ref, atol=1e-10)
    g = names["geom"].index("border1_geom")
    ref = np.array([np.linalg.norm(o.xipos[sb] - o.geom_xpos[g]) for o in oras])
    assert np.allclose(env.distance("sender", "border1_geom"), ref, atol=1e-10)
    data = env.get_data("sender")
    assert np.allclose(data["position"], np.stack([o.xipos[sb] for o in oras]), atol=1e-10)
    assert max(o.ncon for o in oras) > 0
    for e, o in enumerate(oras):
        for c in o.contacts():                       # every contact of the oracle's step is a collision() hit
            assert env.collision(c["geom1"], c["geom2"])[e]
    assert not env.collision("sender_geom", "receiver_geom").any()
    env.close()


# --------------------------------------------------------------------------- reset image, masked and in-launch reset
def test_masked_reset_leaves_the_other_copies_alone(few_build):
    """mjrl_reset(mask): the flagged copies get the reset image (state, warm start, sensordata as mj_forward leaves them
    at the reset state), every other copy keeps every bit -- also its warm start -- and its later trajectory."""
    n_env = 64
    model, packed, h = make("two_agent.xml", n_env)
    model2, _, ref = make("two_agent.xml", n_env)
    rng = np.random.default_rng(5)
    ctrls = rng.uniform(-1, 1, (230, n_env, model.nu))
    for t in range(200):
        for hh in (h, ref):
            hh.set_field("ctrl", ctrls[t])
            hh.step_host(None, 1)
    mask = np.zeros(n_env, np.uint8)
    mask[::5] = 1
    before = {f: h.get_field(f) for f in ("qpos", "qvel", "qacc_warmstart", "sensordata", "ctrl", "timestep")}
    h.reset(mask)
    keep = mask == 0
    for f, old in before.items():
        assert np.array_equal(h.get_field(f)[keep], old[keep]), f
    ora = OracleEnv(packed)
    assert np.array_equal(h.get_field("qpos")[~keep], np.tile(model.qpos0, (int((~keep).sum()), 1)))
    assert np.allclose(h.get_field("qacc_warmstart")[~keep], ora.qacc_warmstart, atol=1e-10)
    assert np.allclose(h.get_field("sensordata")[~keep], ora.sensordata, atol=1e-12)
    assert (h.get_field("timestep")[~keep] == 0).all()
    for t in range(200, 230):
        for hh in (h, ref):
            hh.set_field("ctrl", ctrls[t])
            hh.step_host(None, 1)
    assert np.array_equal(h.get_field("qpos")[keep], ref.get_field("qpos")[keep])
    assert np.array_equal(h.get_field("qvel")[keep], ref.get_field("qvel")[keep])
    # the reset copies follow the oracle from its reset state
    e = 5
    for t in range(200, 230):
        ora.ctrl[:] = ctrls[t, e]
        ora.step()
    assert rel(h.get_field("qpos")[e], ora.qpos) < 1e-9


def test_in_launch_reset_equals_reset_then_step(few_build):
    """mjrl_set_step_reset_mask: flagged copies are reset inside the step launch; bit for bit the result of mjrl_reset on
    those copies followed by the same step.  Also with the mask in device memory for mjrl_reset_device, and at the
    full batch size."""
    import torch
    n_env = 4096
    env_a = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": n_env, "maxSteps": 1024,
                      "environmentDynamics": [__import__("mjrl_amd.dynamics", fromlist=["Language"]).Language]})
    env_b = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": n_env, "maxSteps": 1024,
                      "environmentDynamics": [__import__("mjrl_amd.dynamics", fromlist=["Language"]).Language]})
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.rand((40, 16, 2, 9), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
    acts[..., 8] = (acts[..., 8] + 1) * 1.5
    acts = acts.repeat(1, n_env // 16, 1, 1).contiguous()
    env_a.reset_batched(); env_b.reset_batched()
    out_a = out_b = None
    masks = [torch.zeros(n_env, dtype=torch.uint8, device="cuda") for _ in range(40)]
    rng = np.random.default_rng(2)
    for t in (7, 19, 33):
        masks[t][torch.from_numpy(rng.choice(n_env, 300, replace=False)).cuda()] = 1
    for t in range(40):
        env_a._handle.set_step_reset_mask(masks[t].data_ptr())
        out_a = env_a.step_batched(acts[t])
        if masks[t].any().item():
            env_b._handle.reset_device(masks[t].data_ptr())
        out_b = env_b.step_batched(acts[t])
        torch.cuda.synchronize()
        if t in (7, 8, 19, 33, 39):
            for x, y in zip(out_a, out_b):
                assert torch.equal(x, y), t
            for f in ("qpos", "qvel", "qacc_warmstart", "sensordata", "timestep", "store"):
                assert np.array_equal(env_a._handle.get_field(f), env_b._handle.get_field(f), equal_nan=True), (f, t)
    ts = env_a._handle.get_field("timestep")
    assert ts.max() == 40 and ts.min() == 40 - 33 and set(np.unique(ts)) <= {40, 33, 21, 7}
    env_a.close(); env_b.close()


# --------------------------------------------------------------------------- (iii) full-batch properties
def _full_batch_properties(level, agents, n_env, steps, plugins=()):
    import torch
    cfg = {"xmlPath": levels.level_path(level), "agents": agents, "numEnvs": n_env, "environmentDynamics": list(plugins)}
    env = MuJoCoRL(cfg)
    act_dim = env.action_space(agents[0]).shape[0]
    rng = np.random.default_rng(31)
    base = rng.uniform(-1, 1, (steps, 16, len(agents), act_dim))
    if plugins:
        base[..., 8:] = (base[..., 8:] + 1) * 1.5

    def run(perm):
        env.reset_batched()
        out = None
        for t in range(steps):
            a = torch.from_numpy(np.ascontiguousarray(np.tile(base[t], (n_env // 16, 1, 1))[perm])).cuda()
            out = env.step_batched(a)
        torch.cuda.synchronize()
        return env._handle.get_field("qpos"), out[0].cpu().numpy(), out[1].cpu().numpy()

    ident = np.arange(n_env)
    q1, o1, r1 = run(ident)
    assert np.isfinite(q1).all() and np.isfinite(o1).all()
    assert np.array_equal(q1[:16], q1[16:32]) and np.array_equal(q1[:16], q1[-16:])
    assert np.array_equal(o1[:16], o1[-16:])
    q2, o2, r2 = run(ident)
    assert np.array_equal(q1, q2) and np.array_equal(o1, o2) and np.array_equal(r1, r2)
    perm = rng.permutation(n_env)
    q3, o3, r3 = run(perm)
    assert np.array_equal(q3, q1[perm]) and np.array_equal(o3, o1[perm])
    assert env._handle.cap_overflows() == (0, 0)
    return env, base, q1, o1


def test_full_batch_properties_four_agent_arena():
    """Config 4 at its per-GPU size: 4096 copies of the 4-agent arena (38 KiB LDS image, 4 copies per CU, the
    two-positions-per-lane schedule solver)."""
    steps = 260
    env, base, q1, o1 = _full_batch_properties("four_agent.xml", FOUR, 4096, steps)
    oras = [OracleEnv(env._blob) for _ in range(3)]
    for t in range(steps):
        for e, o in enumerate(oras):
            for k, a in enumerate(FOUR):
                o.ctrl[env.agents_action_index[a]] = base[t, e, k]
            o.step()
    assert max(o.ncon for o in oras) > 0
    assert rel(q1[:3], np.stack([o.qpos for o in oras])) < 1e-9
    env.close()


def test_full_batch_properties_fused_config_three():
    """Config 3 at its size: 4096 copies of the 2-agent level with the Language channel fused into the launch."""
    from mjrl_amd.dynamics import Language
    steps = 230
    env, base, q1, o1 = _full_batch_properties("two_agent.xml", AGENTS, 4096, steps, plugins=[Language])
    assert o1.shape == (4096, 2, 60)
    # what each agent hears is the other's utterance of the same step (receiver) / of the step before (sender)
    assert np.array_equal(o1[:16, 1, 59], np.trunc(base[-1, :, 0, 8]))
    assert np.array_equal(o1[:16, 0, 59], np.trunc(base[-2, :, 1, 8]))
    oras = [OracleEnv(env._blob) for _ in range(3)]
    for t in range(steps):
        for e, o in enumerate(oras):
            for k, a in enumerate(AGENTS):
                o.ctrl[env.agents_action_index[a]] = base[t, e, k, :8]
            o.step()
    assert rel(q1[:3], np.stack([o.qpos for o in oras])) < 1e-9
    for e, o in enumerate(oras):
        assert np.allclose(o1[e, 0, :59], np.concatenate([o.sensordata[[0]], o.qpos, o.qvel]), atol=1e-9)
    env.close()


# --------------------------------------------------------------------------- (iv) cameras at config 5's size
def test_camera_batch_properties_at_512_copies(few_build):
    """Config 5's batch: 512 copies x 2 cameras x 64x64x3.  Copies in the same state render the same bytes wherever
    they sit in the batch, runs repeat bit for bit, and sampled copies match the oracle's ray caster."""
    n_env = 512
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": n_env, "agentCameras": True})
    env.reset()
    rng = np.random.default_rng(13)
    base = rng.uniform(-1, 1, (60, 8, 2, 8))
    for t in range(60):
        act = np.tile(base[t], (n_env // 8, 1, 1))
        env.step({a: act[:, k] for k, a in enumerate(AGENTS)})
    # (this test moves states between copies by hand: images of the CURRENT qpos, i.e. without the scene cache that
    # agentCameras turned on -- the cached frames of a step are covered by test_render_draws_the_frames_of_the_last_forward_pass)
    env._handle.set_scene_cache(False)
    img1 = env._handle.render(64, 64)
    assert img1.shape == (n_env, 2, 64, 64, 3) and img1.dtype == np.uint8
    assert np.array_equal(img1[:8], img1[8:16]) and np.array_equal(img1[:8], img1[-8:])
    assert np.array_equal(img1, env._handle.render(64, 64))
    assert len({img1[e].tobytes() for e in range(8)}) > 1              # different states, different images
    perm = rng.permutation(n_env)
    qpos = env._handle.get_field("qpos")
    env._handle.set_field("qpos", qpos[perm])
    assert np.array_equal(env._handle.render(64, 64), img1[perm])
    o = OracleEnv(env._blob)
    for e in (0, 3, 7):
        o.qpos[:] = qpos[e]
        o.forward()
        for cam in range(2):
            ref = o.render(cam, 64, 64).reshape(64, 64, 3).astype(int)
            differ = np.abs(ref - img1[e, cam].astype(int)).max(axis=-1) > 0
            assert differ.mean() < 0.004
    env.close()


# --------------------------------------------------------------------------- multi-level re-initialisation (ADVICE)
def test_level_switch_on_reset_keeps_the_device_configuration(tmp_path):
    """An ``xmlPath`` list makes reset() pick a level at random and re-create model and data
    (mujoco_parent.py:351-356).  The new device handle must get what the old one had: tables, truncation horizon,
    the fused plugin program (Language slot, rewards, dones) -- the reference keeps its Python plugin loop."""
    from mjrl_amd.dynamics import Language, TargetDistanceReward
    text = open(levels.level_path("two_agent.xml")).read()
    paths = []
    # two levels that differ in STRUCTURE (a platform stands elsewhere): not colour variants of one model, so reset()
    # switches the whole batch like the reference does and re-creates the device state
    for k, where in enumerate(("7.02852 -2.071592 0.4710507", "6.2 -1.5 0.4710507")):
        p = tmp_path / f"level{k}.xml"
        p.write_text(text.replace('pos="7.02852 -2.071592 0.4710507"', f'pos="{where}"'))
        paths.append(str(p))
    random.seed(0)
    env = MuJoCoRL({"xmlPath": paths, "agents": AGENTS, "numEnvs": 4, "maxSteps": 3, "environmentDynamics": [Language],
                    "rewardFunctions": [TargetDistanceReward("reference", mode="negative")]})
    assert env._program is not None and env._variants is None
    seen, handles = set(), []
    for episode in range(12):
        env.reset()
        seen.add(env.xml_path)
        handle = env._handle
        if not any(handle is h for h in handles):
            handles.append(handle)          # (kept referenced, so that a new handle is a new object)
        assert handle.size("n_slot") == len(env._program.slots) and handle.size("obs_dim") == 60
        for call in range(5):
            act = {a: np.concatenate([np.zeros((4, 8)), np.full((4, 1), 2.0 if a == "sender" else 1.0)], axis=1) for a in AGENTS}
            obs, rew, term, trunc, info = env.step(act)
            assert obs["receiver"].shape == (4, 60) and (obs["receiver"][:, 59] == 2).all()
            assert (rew["sender"] < 0).all()
            assert trunc["sender"].all() == (call >= 3)
        if len(seen) == 2 and episode >= 3:
            break
    assert len(seen) == 2 and len(handles) >= 2, "the level never switched"
    env.close()


# --------------------------------------------------------------------------- device-buffer validation (ADVICE)
def test_step_batched_rejects_buffers_that_would_fault():
    import torch
    env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 8})
    env.reset_batched()
    good = torch.zeros((8, 2, 8), dtype=torch.float64, device="cuda")
    env.step_batched(good)
    with pytest.raises(Exception, match="shape"):
        env.step_batched(torch.zeros((4, 2, 8), dtype=torch.float64, device="cuda"))
    with pytest.raises(Exception, match="shape"):
        env.step_batched(torch.zeros((8, 2, 5), dtype=torch.float64, device="cuda"))
    with pytest.raises(Exception, match="contiguous"):
        env.step_batched(torch.zeros((8, 2, 8), dtype=torch.float32, device="cuda"))
    with pytest.raises(Exception, match="contiguous"):
        env.step_batched(torch.zeros((8, 2, 16), dtype=torch.float64, device="cuda")[:, :, ::2])
    with pytest.raises(Exception, match="shape"):
        env.step_batched(good, obs=torch.zeros((8, 2, 30), dtype=torch.float64, device="cuda"))
    with pytest.raises(Exception, match="contiguous"):
        env.step_batched(good, term=torch.zeros((8, 2), dtype=torch.float64, device="cuda"))
    with pytest.raises(Exception, match="lives on"):
        env.step_batched(good, reward=torch.zeros((8, 2), dtype=torch.float64))
    env.close()


# --------------------------------------------------------------------------- info-JSON tags, target / pick-up ops
INFO_JSON = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "two_agent_info.json")


def test_tag_ops_fused_equal_the_host_plugin_loop(few_build):
    """the info JSON's tags as device tables.  PickUpDynamic (Testing/Pick_Up_Dynamic.py:15-41) with a
    distance-decrease reward and a done on the agent's current target, once as ops of the step kernel and once through
    the host plugin loop on the same GPU physics; the host side goes through filter_by_tag / get_data."""
    from mjrl_amd.dynamics import PickUpDynamic, TargetDistanceReward, TargetReached

    class Pick(PickUpDynamic):
        threshold, seed = 5.0, 99

    def make_env(fused):
        return MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "infoJson": INFO_JSON, "agents": AGENTS, "numEnvs": 6,
                         "environmentDynamics": [Pick], "fusedPlugins": fused, "firstEnvId": 4096,
                         "rewardFunctions": [TargetDistanceReward(current_target_of="target", mode="delta", scale=3.0)],
                         "doneFunctions": [TargetReached(current_target_of="target", threshold=3.0)]})
    fused, host = make_env(True), make_env(False)
    assert fused._program is not None and fused._program.tags == ["target"] and host._program is None
    assert fused._handle.size("n_tag") == 1 and fused._handle.size("env_base") == 4096
    assert fused.observation_space("sender").shape == (59 + 4,) and fused.action_space("sender").shape == (8,)
    found = host.filter_by_tag("target")
    assert [d["name"] for d in found] == ["reference", "choice_1", "choice_2"] and found[0]["class"] == "Cube"
    np.random.seed(0); fused.reset()
    np.random.seed(0); host.reset()
    rng = np.random.default_rng(3)
    toggled = 0
    for step in range(40):
        action = {a: rng.uniform(-1, 1, (6, 8)) for a in AGENTS}
        f, h = fused.step(action), host.step(action)
        for a in AGENTS:
            assert np.allclose(f[0][a], h[0][a], atol=1e-12), (step, a)
            assert np.array_equal(f[0][a][:, 62], h[0][a][:, 62])                 # inventory
            assert np.allclose(f[1][a], h[1][a], atol=1e-10)
            assert np.array_equal(f[2][a], h[2][a])
            toggled += int((f[1][a] >= 1.0 - 1e-9).sum())
        assert np.array_equal(f[2]["__all__"], h[2]["__all__"])
    assert toggled > 0
    store = fused.device_store
    for a in AGENTS:
        assert np.array_equal(store[a]["current_target"], np.asarray(host.data_store[a]["current_target"], np.float64))
        assert np.array_equal(store[a]["inventory"], host.data_store[a]["inventory"])
    # copies draw different targets (keyed on the global copy id)
    assert len(set(store["receiver"]["current_target"].tolist()) | set(store["sender"]["current_target"].tolist())) > 1
    fused.close(); host.close()


# --------------------------------------------------------------------------- per-copy level variants
def test_per_copy_level_variants(tmp_path):
    """an xmlPath list of levels that differ in colours only (Testing/levels/Model2-10.xml) is one model
    with per-copy colour variants drawn at every reset of a copy (mujoco_parent.py:351-356).  A copy's physics does not
    depend on its variant; its camera pixels do."""
    from mjrl_amd.dynamics import mix64, pick_of
    text = open(levels.level_path("two_agent.xml")).read()
    paths = []
    for k, colour in enumerate(("0 .9 0 1", "0.9 0 0 1", "0 0 0.9 1")):
        p = tmp_path / f"Model{k + 2}.xml"
        p.write_text(text.replace('rgba="0 .9 0 1"', f'rgba="{colour}"'))
        paths.append(str(p))
    n_env = 48
    env = MuJoCoRL({"xmlPath": paths, "agents": AGENTS, "numEnvs": n_env, "agentCameras": True, "variantSeed": 5})
    assert env._variants is not None and env._handle.size("n_variant") == 3
    handle = env._handle
    env.reset()
    assert env._handle is handle                                    # no model switch, no new device state
    ids = env.variant_ids()
    assert np.array_equal(ids, pick_of(mix64(5, np.arange(n_env), 0, 1, 2), 3)) and set(ids) == {0, 1, 2}
    assert env.xml_path == paths[ids[0]]
    act = np.random.default_rng(0).uniform(-1, 1, (30, 2, 8))
    for t in range(30):
        env.step({a: np.tile(act[t, k], (n_env, 1)) for k, a in enumerate(AGENTS)})
    qpos = env._handle.get_field("qpos")
    assert np.array_equal(qpos, np.tile(qpos[0], (n_env, 1)))       # the physics does not know the variant
    images = env._handle.render(64, 64)
    by_variant = {v: images[np.flatnonzero(ids == v)[0]] for v in (0, 1, 2)}
    for e in range(n_env):
        assert np.array_equal(images[e], by_variant[ids[e]])        # same state + same variant -> same bytes
    assert not np.array_equal(by_variant[0], by_variant[1]) and not np.array_equal(by_variant[1], by_variant[2])
    # a masked reset starts a new episode for the flagged copies only: they draw again
    mask = np.zeros(n_env, np.uint8)
    mask[::3] = 1
    env._handle.reset(mask)
    ids2 = env.variant_ids()
    assert np.array_equal(ids2[mask == 0], ids[mask == 0])
    expect = pick_of(mix64(5, np.arange(n_env), 0, 2, 2), 3)
    assert np.array_equal(ids2[mask == 1], expect[mask == 1])
    assert np.array_equal(env._handle.get_field("episode"), 1 + mask.astype(np.int32))
    # the in-launch reset draws the same way
    import torch
    m = torch.from_numpy(mask).cuda()
    env._handle.set_step_reset_mask(m.data_ptr())
    env.step_batched(torch.zeros((n_env, 2, 8), dtype=torch.float64, device="cuda"))
    env._handle.set_step_reset_mask(None)
    torch.cuda.synchronize()
    expect3 = pick_of(mix64(5, np.arange(n_env), 0, 3, 2), 3)
    ids3 = env.variant_ids()
    assert np.array_equal(ids3[mask == 1], expect3[mask == 1]) and np.array_equal(ids3[mask == 0], ids[mask == 0])
    env.close()
    # levels that differ in structure keep the reference's whole-batch switch
    other = MuJoCoRL({"xmlPath": [levels.level_path("two_agent.xml"), levels.level_path("two_agent_2sensors.xml")],
                      "agents": AGENTS, "numEnvs": 2})
    assert other._variants is None
    other.close()


# (the vector-env adapter against the oracle through autoresets: tests/test_gpu_parity_r3.py, round 3 moved it onto the fast path)


# --------------------------------------------------------------------------- box-box contacts
def test_box_box_contacts_on_the_device():
    """The free BOX agent of the reference's sensor configurations (Testing/sensor_test.py:20, Testing/sensor_levels/
    Model1.xml) held at an arena wall, and pairs of free boxes in random poses: contact lists, solver sweeps and
    trajectories against the oracle (the routine's known answers are in tests/test_box_box.py)."""
    from tests.test_box_box import random_pose_xml
    env = MuJoCoRL({"xmlPath": levels.level_path("sensor_touch.xml"), "agents": ["receiver"], "numEnvs": 4, "freeJoint": True})
    env.reset()
    qpos = env._handle.get_field("qpos")
    qpos[:, 1] = 3.6
    env._handle.set_field("qpos", qpos)
    oras = [OracleEnv(env._blob) for _ in range(4)]
    for o in oras:
        o.qpos[1] = 3.6
    drive = np.array([[0.0, 1.0, 0.0], [0.3, 1.0, 0.2], [-0.4, 0.9, -0.5], [0.0, 0.6, 1.0]])
    for _ in range(450):
        env.step({"receiver": drive})
        for e, o in enumerate(oras):
            o.qvel[[0, 1, 5]] = drive[e]
            o.step()
    names = env._compiled.names["geom"]
    wall, box = names.index("border2_geom"), names.index("receiver_geom")
    assert all(any((c["geom1"], c["geom2"]) == (wall, box) for c in o.contacts()) for o in oras[:3])
    assert rel(env._handle.get_field("qpos"), np.stack([o.qpos for o in oras])) < 1e-9
    stats = env._handle.get_field("solver_stats")
    assert np.array_equal(stats[:, 0], [o.ncon for o in oras]) and np.array_equal(stats[:, 2], [o.niter for o in oras])
    assert (env._handle.get_field("qpos")[:3, 1] + 0.5 < 4.738263 - 0.25 + 0.2).all()       # held at the wall
    env.close()
    rng = np.random.default_rng(17)
    touched = 0
    for trial in range(12):
        model = mjcf.compile_mjcf_string(random_pose_xml(rng))
        packed = blob.pack(model)
        h = _capi.Handle(packed, 3)
        h.reset()
        ora = OracleEnv(packed)
        for _ in range(6):
            h.step_host(None, 1)
            ora.step()
            stats = h.get_field("solver_stats")
            assert (stats[:, 0] == ora.ncon).all() and (stats[:, 2] == ora.niter).all(), trial
            touched += ora.ncon > 0
        assert rel(h.get_field("qpos"), np.tile(ora.qpos, (3, 1))) < 1e-10
        h.close()
    assert touched > 10


# --------------------------------------------------------------------------- Runge-Kutta integrator
def test_rk4_level_through_the_c_abi(few_build):
    """benchmarking/levels/Ant.xml (RK4, dt 0.01): four launches per physics frame, against the oracle's RK4 step; through
    the env class with skipFrames 2."""
    n_env = 5
    env = MuJoCoRL({"xmlPath": levels.level_path("ant.xml"), "agents": ["torso"], "numEnvs": n_env, "skipFrames": 2})
    assert env.agents_action_index == {"torso": [2, 3, 4, 5, 6, 7, 0, 1]} and env.observation_space("torso").shape == (29,)
    env.reset()
    oras = [OracleEnv(env._blob) for _ in range(n_env)]
    rng = np.random.default_rng(4)
    for step in range(80):
        act = rng.uniform(-1, 1, (n_env, 8))
        obs, rew, term, trunc, info = env.step({"torso": act})
        for e, o in enumerate(oras):
            o.ctrl[env.agents_action_index["torso"]] = act[e]
            o.step(2)
    assert max(o.ncon for o in oras) > 0
    assert rel(env._handle.get_field("qpos"), np.stack([o.qpos for o in oras])) < 1e-9
    assert rel(env._handle.get_field("qvel"), np.stack([o.qvel for o in oras])) < 1e-8
    assert np.allclose(obs["torso"], np.stack([np.concatenate([o.qpos, o.qvel]) for o in oras]), atol=1e-8)
    assert np.array_equal(env._handle.get_field("timestep"), np.full(n_env, 80, np.int32))
    stats = env._handle.get_field("solver_stats")
    assert np.array_equal(stats[:, 0], [o.ncon for o in oras]) and np.array_equal(stats[:, 2], [o.niter for o in oras])
    env.close()


# --------------------------------------------------------------------------- random scenes on the hardware
def test_random_scenes_on_the_device(few_build):
    """The scenes of tests/test_fuzz_scenes.py (mixed geoms, one to four trees, hinge / slide / free joints) through the
    C-ABI with the generic kernel (one code object for every shape), against the oracle: contact, row and sweep counts
    every 20 steps, trajectories at the end."""
    from tests.test_fuzz_scenes import random_scene
    for seed in list(range(0, 8)) + list(range(3000, 3008)):
        rng = np.random.default_rng(seed)
        model = mjcf.compile_mjcf_string(random_scene(rng), nconmax=24, njmax=120)
        packed = blob.pack(model)
        h = _capi.Handle(packed, 3, specialize=False)
        h.reset()
        ora = OracleEnv(packed)
        qvel = h.get_field("qvel")
        for j in range(model.njnt):
            if model.jnt_type[j] == mjcf.JNT_FREE:
                qa, da = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
                ora.qvel[da:da + 2] = -2.0 * ora.qpos[qa:qa + 2]
                qvel[:, da:da + 2] = -2.0 * model.qpos0[qa:qa + 2]
        h.set_field("qvel", qvel)
        for step in range(240):
            h.step_host(None, 1)
            ora.step()
            if step % 20 == 19:
                stats = h.get_field("solver_stats")
                assert (stats[:, 0] == ora.ncon).all() and (stats[:, 1] == ora.nefc).all() and (stats[:, 2] == ora.niter).all(), (seed, step)
        assert rel(h.get_field("qpos"), np.tile(ora.qpos, (3, 1))) < 1e-8, seed
        h.close()


@pytest.mark.gpu
def test_pinned_host_buffers_give_what_the_copying_host_path_gives():
    """``step_batched(numpy actions)`` without output arrays runs on the handle's pinned host buffers (the kernel reads
    the action rows and writes the result rows over PCIe itself, ``mjrl_step_pinned``) and returns views of them; with
    output arrays it is ``mjrl_step_host`` (copies).  Two envs stepped the two ways agree bit for bit, fused Language
    channel, truncation flags and in-place reuse of the views included."""
    from mjrl_amd.dynamics import Language
    cfg = {"xmlPath": levels.level_path("two_agent.xml"), "agents": AGENTS, "numEnvs": 96, "maxSteps": 12,
           "environmentDynamics": [Language]}
    a, b = MuJoCoRL(dict(cfg)), MuJoCoRL(dict(cfg))
    a.reset(); b.reset()
    n_agent, obs_dim = 2, a._handle.size("obs_dim")
    obs = np.zeros((96, n_agent, obs_dim)); rew = np.zeros((96, n_agent))
    term = np.zeros((96, n_agent), np.uint8); trunc = np.zeros((96, n_agent), np.uint8)
    rng = np.random.default_rng(4)
    first = None
    for t in range(16):
        act = rng.uniform(-1, 1, (96, n_agent, 9))
        out = a.step_batched(act)
        b.step_batched(act, obs, rew, term, trunc)
        for x, y in zip(out, (obs, rew, term, trunc)):
            assert np.array_equal(x, y), t
        first = out[0] if first is None else first
        assert out[0] is first or np.shares_memory(out[0], first)      # the same pinned buffer every step
    assert trunc.any() and np.abs(obs).sum() > 0


@pytest.mark.gpu
def test_every_copy_is_stepped_exactly_once_for_odd_batch_sizes():
    """The longest-first dispatch hands workgroup w the copy of the w-th set bit of the previous launch's work buckets
    (bit rows of (n_env + 31) / 32 words, two words per lane and block of 128).  Batches that are not a multiple of 32 or
    64, a single copy, and one with more than one block of words: after every launch every copy's step counter has moved
    by exactly one (a copy skipped or stepped twice would show), the lost-workgroup counter stays zero, and copies at the
    ends and in the middle follow the oracle."""
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    packed = blob.pack(model)
    for n_env in (1, 33, 130, 5000):
        h = _capi.Handle(packed, n_env)
        h.reset()
        rng = np.random.default_rng(n_env)
        watch = sorted({0, n_env // 2, n_env - 1})
        oras = {e: OracleEnv(packed) for e in watch}
        z = rng.uniform(0.12, 0.9, n_env)                       # different heights: different work, so several buckets fill
        q = np.tile(model.qpos0, (n_env, 1))
        q[:, 2] = z
        h.set_field("qpos", q)
        for e, o in oras.items():
            o.qpos[:] = q[e]
        for t in range(6):
            ctrl = rng.uniform(-1, 1, (n_env, model.nu))
            h.set_field("ctrl", ctrl)
            h.step_device(None, 0, 1)
            assert np.array_equal(h.get_field("timestep"), np.full(n_env, t + 1)), (n_env, t)
            for e, o in oras.items():
                o.ctrl[:] = ctrl[e]
                o.step()
        assert h.cap_overflows() == (0, 0)                       # (raises if a workgroup found no copy)
        got = h.get_field("qpos")
        for e, o in oras.items():
            assert np.allclose(got[e], o.qpos, rtol=0, atol=1e-10), (n_env, e)
        h.close()
