"""The device step code (csrc/mjrl_step.h, compiled for the CPU with 64 lanes run as fibers -- tests/emu) against
the CPU oracle on identical models and action sequences.  This checks the kernel's logic, lane mappings and
barrier placement without a GPU; the -m gpu tests repeat the comparison through the real C-ABI on an MI355X.
Tolerance: both sides are fp64 with the same operation order up to reduction shape, so trajectories agree to
1e-9 absolute over hundreds of contact-rich steps (observed: 1e-12)."""
import numpy as np
import pytest

from mjrl_amd import blob, levels, mjcf
from oracle.oracle import OracleEnv
from tests.emu.emu import EmuEnv


def pair(level, **kw):
    model = mjcf.compile_mjcf(levels.level_path(level), **kw)
    packed = blob.pack(model)
    ora, emu = OracleEnv(packed), EmuEnv(model, packed)
    emu.step(forward_only=True)      # mj_forward after the reset, like mjrl_reset does
    return model, ora, emu


def test_reset_forward_pass_matches():
    model, ora, emu = pair("two_agent.xml")
    assert np.allclose(emu.warm, ora.qacc_warmstart, atol=1e-10)
    assert np.allclose(emu.sens[:2], ora.sensordata, atol=1e-12)
    assert np.array_equal(emu.qpos, model.qpos0)


def test_intermediate_quantities_match_after_contacts_appear():
    model, ora, emu = pair("two_agent.xml")
    rng = np.random.default_rng(11)
    img = None
    for _ in range(230):
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl
        emu.ctrl[:] = ctrl
        img = emu.step()
        ora.step()
        assert (img.nefc, img.ncon, img.niter) == (ora.nefc, ora.ncon, ora.niter)
    assert ora.ncon > 0
    assert np.allclose(img.M_from_factor(), ora.qM, rtol=1e-10, atol=1e-10)
    for name, ref in (("xpos", ora.xpos), ("xquat", ora.xquat), ("cdof", ora.cdof),
                      ("cvel", ora.cvel), ("bias", ora.qfrc_bias), ("qaccs", ora.qacc_smooth)):
        assert np.allclose(img.region(name), ref, rtol=0, atol=1e-9), name
    # (geom frames are kept for the collision stage only; from the dumped body frames they follow as the kernel forms them)
    xpos, xquat = img.region("xpos"), img.region("xquat")
    gpos = np.stack([xpos[b] + mjcf.quat_to_mat(xquat[b]) @ model.geom_pos[g] for g, b in enumerate(model.geom_bodyid)])
    gmat = np.stack([mjcf.quat_to_mat(mjcf.quat_mul(xquat[b], model.geom_quat[g])).reshape(9)
                     for g, b in enumerate(model.geom_bodyid)])
    assert np.allclose(gpos, ora.geom_xpos, atol=1e-9) and np.allclose(gmat, ora.geom_xmat, atol=1e-9)
    assert np.allclose(img.region("qacc"), ora.qacc, rtol=1e-9, atol=1e-7)
    assert np.allclose(img.region("qfc"), ora.qfrc_constraint, rtol=1e-9, atol=1e-7)
    cons = ora.contacts()
    assert [(c["geom1"], c["geom2"]) for c in cons] == [tuple(x) for x in img.contact_geoms()]
    assert np.allclose(img.region("con")[:len(cons), 0], [c["dist"] for c in cons], atol=1e-12)
    assert np.allclose(img.region("row")[:ora.nefc, 2], ora.efc_force[:ora.nefc], rtol=1e-9, atol=1e-7)


def test_constraint_rows_match_before_projection():
    """Raw Jacobian rows, regularisation and bias of one contact-rich step (a stage-1 dump leaves J unprojected, so
    that launch is for inspection only and ends the comparison)."""
    model, ora, emu = pair("two_agent.xml")
    rng = np.random.default_rng(5)
    for k in range(500):
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl
        emu.ctrl[:] = ctrl
        emu.step()
        ora.step()
        if k > 150 and ora.ncon >= 2:
            break
    ctrl = rng.uniform(-1, 1, model.nu)
    ora.ctrl[:] = ctrl
    emu.ctrl[:] = ctrl
    img = emu.step(dbg_stage=1)
    ora.step()
    n = ora.nefc
    assert n >= 8 and img.nefc == n
    assert np.allclose(img.J()[:n], ora.efc_J[:n], atol=1e-12)
    rows = img.region("row")[:n]
    assert np.allclose(rows[:, 0], ora.efc_R[:n], rtol=1e-12)
    assert np.allclose(rows[:, 1], ora.efc_b[:n], rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("level,steps", [("two_agent.xml", 400), ("single_agent.xml", 300), ("two_agent_3sensors.xml", 250),
                                         ("four_agent.xml", 260)])
def test_trajectory_parity(level, steps, emu_few):
    model, ora, emu = pair(level)
    rng = np.random.default_rng(3)
    saw_contact = False
    for _ in range(steps):
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl
        emu.ctrl[:model.nu] = ctrl
        img = emu.step()
        ora.step()
        saw_contact |= ora.ncon > 0
        assert (img.nefc, img.ncon) == (ora.nefc, ora.ncon)
    assert saw_contact
    assert np.allclose(emu.qpos, ora.qpos, rtol=0, atol=1e-9)
    assert np.allclose(emu.qvel, ora.qvel, rtol=0, atol=1e-8)
    assert np.allclose(emu.sens[:model.nsensordata], ora.sensordata, rtol=0, atol=1e-7)


@pytest.mark.parametrize("kind", ["touch", "accelerometer", "rangefinder", "framexaxis"])
def test_sensor_levels(kind):
    """Free box on the floor with one site sensor (Testing/sensor_levels): plane-box contacts, every sensor type."""
    model, ora, emu = pair(f"sensor_{kind}.xml")
    for _ in range(120):
        img = emu.step()
        ora.step()
        assert (img.nefc, img.ncon) == (ora.nefc, ora.ncon)
    assert ora.ncon == 4                     # the four lower corners of the box
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-10)
    assert np.allclose(emu.sens[:model.nsensordata], ora.sensordata, atol=1e-8)


def test_skip_frames_equals_repeated_steps():
    model, ora, emu = pair("two_agent.xml")
    ctrl = np.linspace(-1, 1, model.nu)
    ora.ctrl[:] = ctrl
    emu.ctrl[:] = ctrl
    emu.step(skip_frames=5)
    ora.step(5)
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-12)


def test_action_scatter_and_observation_gather():
    model, ora, emu = pair("two_agent.xml")
    scatter = np.array([[2, 3, 4, 5, 6, 7, 0, 1, -1], [10, 11, 12, 13, 14, 15, 8, 9, -1]], np.int32)   # 9th slot: plugin action
    gather = np.full((2, 59), -1, np.int32)
    for a in range(2):
        gather[a, 0] = (0 << 24) | a
        gather[a, 1:31] = (1 << 24) | np.arange(30)
        gather[a, 31:59] = (2 << 24) | np.arange(28)
    rng = np.random.default_rng(0)
    obs = np.zeros((2, 59))
    for _ in range(30):
        actions = rng.uniform(-1, 1, (2, 9))
        emu.step(actions=actions, scatter=scatter, n_agent=2, gather=gather, obs=obs)
        for a in range(2):
            ora.ctrl[scatter[a, :8]] = actions[a, :8]
        ora.step()
    assert np.allclose(emu.ctrl[:16], ora.ctrl)
    for a in range(2):
        expect = np.concatenate([ora.sensordata[[a]], ora.qpos, ora.qvel])
        assert np.allclose(obs[a], expect, atol=1e-10)
    assert emu.timestep[0] == 30


def test_free_joint_mode_overwrites_qvel():
    """freeJoint=True writes the action into qvel[dof, dof+1, dof+5] before the physics (mujoco_parent.py:325)."""
    model, ora, emu = pair("two_agent.xml")
    scatter = np.array([[0, 1, 5], [14, 15, 19]], np.int32)
    actions = np.array([[0.5, -0.25, 1.0], [-1.0, 0.75, 0.1]])
    for _ in range(10):
        emu.step(actions=actions, scatter=scatter, n_agent=2, scatter_mode=1)
        for a in range(2):
            ora.qvel[scatter[a]] = actions[a]
        ora.step()
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-12) and np.allclose(emu.qvel, ora.qvel, atol=1e-11)


def test_contact_cap_drops_in_order_and_flags():
    model, ora, emu = pair("sensor_touch.xml", nconmax=2, njmax=8)
    for _ in range(40):
        img = emu.step()
        ora.step()
    assert img.ncon == ora.ncon == 2 and img.warn & 1 and ora.warnings & 1
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-10)


def test_fused_plugin_ops_follow_the_reference_loop_order(emu_few):
    """Language channel + distance reward / done as ops of the step kernel, against a direct transcription of the
    reference's plugin loop (dynamic-major, agent-minor, each call seeing the earlier calls' data-store writes;
    README.md:109-136, mujoco_rl.py:215-241, 276-286)."""
    model, ora, emu = pair("two_agent.xml")
    scatter = np.array([[2, 3, 4, 5, 6, 7, 0, 1, -1], [10, 11, 12, 13, 14, 15, 8, 9, -1]], np.int32)
    gather = np.full((2, 60), -1, np.int32)
    for a in range(2):
        gather[a, 0] = a
        gather[a, 1:31] = (1 << 24) | np.arange(30)
        gather[a, 31:59] = (2 << 24) | np.arange(28)
        gather[a, 59] = -2
    bodies = np.array([model.names["body"].index("sender"), model.names["body"].index("receiver")], np.int32)
    target = model.names["body"].index("reference")
    prog_i = np.array([[1, 8, 0, 0, 0, 0, 0, 0], [2, 0, target, 1, 1, 0, 0, 0], [3, 0, target, 0, 0, 0, 0, 0]], np.int32)
    prog_f = np.array([[0, 0, 0, 0], [2.0, 0, 0, 0], [3.4, 0, 0, 0]], np.float64)
    program = dict(prog_i=prog_i, prog_f=prog_f, n_slot=2, agent_body=bodies, agent_obs_len=np.array([59, 59], np.int32),
                   store=np.full((2, 2), np.nan), reward=np.zeros(2), term=np.zeros(2, np.uint8), trunc=np.zeros(2, np.uint8))
    obs = np.zeros((2, 60))
    rng = np.random.default_rng(8)
    store = [dict(), dict()]
    for step in range(25):
        actions = np.concatenate([rng.uniform(-1, 1, (2, 8)), rng.uniform(0, 3, (2, 1))], axis=1)
        emu.step(actions=actions, scatter=scatter, n_agent=2, gather=gather, obs=obs, program=program, max_steps=20)
        for a in range(2):
            ora.ctrl[scatter[a, :8]] = actions[a, :8]
        ora.step()
        # the reference loop, written out
        heard, rewards, dones = [0, 0], [0.0, 0.0], [False, False]
        for a in range(2):
            store[a]["utterance"] = int(actions[a, 8])
            heard[a] = store[1 - a].get("utterance", 0)
        for a in range(2):
            dist = np.linalg.norm(ora.xipos[bodies[a]] - ora.xipos[target])
            if "distance" in store[a]:
                rewards[a] += 2.0 * (store[a]["distance"] - dist)
            store[a]["distance"] = dist
        for a in range(2):
            dones[a] = np.linalg.norm(ora.xipos[bodies[a]] - ora.xipos[target]) < 3.4
        assert [obs[0, 59], obs[1, 59]] == heard
        assert np.allclose(program["reward"], rewards, atol=1e-10)
        assert list(program["term"].astype(bool)) == dones
        assert program["trunc"].all() == (step >= 20)
    assert any(dones) and not all(dones)      # the threshold separates the two agents


def _rows_couple_two_chains(model, ora):
    """(rows between two trees, rows between two moving bodies of one tree) among the oracle's current contacts"""
    tree, gbody = model.body_treeid, model.geom_bodyid
    cross = same = 0
    for c in ora.contacts():
        t1, t2 = int(tree[gbody[c["geom1"]]]), int(tree[gbody[c["geom2"]]])
        if t1 >= 0 and t2 >= 0:
            cross += t1 != t2
            same += t1 == t2
    return cross, same


def test_rows_between_two_moving_bodies():
    """Contacts whose two bodies both carry dofs build their rows from two ancestor chains: agent against agent
    (two kinematic trees, the solver's serial fallback) and leg against leg of one agent (chains that share the
    torso dofs).  Both states are posed by hand: random play rarely produces them."""
    model, ora, emu = pair("two_agent.xml")
    free = [j for j in range(model.njnt) if model.jnt_type[j] == 0]
    assert len(free) == 2
    a0, a1 = (int(model.jnt_qposadr[j]) for j in free)
    # agent 1 dropped onto agent 0
    q = model.qpos0.copy()
    q[a1:a1 + 3] = q[a0:a0 + 3] + np.array([0.15, 0.1, 0.55])
    ora.qpos[:] = q; emu.qpos[:] = q
    seen_cross = 0
    for k in range(60):
        img = emu.step(); ora.step()
        cross, _ = _rows_couple_two_chains(model, ora)
        seen_cross += cross
        assert (img.nefc, img.ncon, img.niter) == (ora.nefc, ora.ncon, ora.niter), k
    assert seen_cross > 0
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-9) and np.allclose(emu.qvel, ora.qvel, atol=1e-8)
    # legs of one agent folded into each other (hinge angles outside their ranges: the limit rows push back)
    model, ora, emu = pair("two_agent.xml")
    rng = np.random.default_rng(3)
    hinge_q = [int(model.jnt_qposadr[j]) for j in range(model.njnt) if model.jnt_type[j] == 3]
    found = None
    for trial in range(200):
        q = model.qpos0.copy()
        q[hinge_q] = rng.uniform(-2.5, 2.5, len(hinge_q))
        q[a0 + 2] += 1.0; q[a1 + 2] += 1.0          # in the air: only self-contacts
        ora.qpos[:] = q; ora.qvel[:] = 0
        ora.forward()
        if _rows_couple_two_chains(model, ora)[1] > 0:
            found = q
            break
    assert found is not None
    ora.qpos[:] = found; emu.qpos[:] = found
    img = emu.step(dbg_stage=1); ora.step()
    n = ora.nefc
    assert img.nefc == n and np.allclose(img.J()[:n], ora.efc_J[:n], atol=1e-12)      # raw two-chain rows
    model, ora, emu = pair("two_agent.xml")
    ora.qpos[:] = found; emu.qpos[:] = found
    seen_same = 0
    for k in range(40):
        img = emu.step(); ora.step()
        seen_same += _rows_couple_two_chains(model, ora)[1]
        assert (img.nefc, img.ncon, img.niter) == (ora.nefc, ora.ncon, ora.niter), k
    assert seen_same > 0
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-9) and np.allclose(emu.qvel, ora.qvel, atol=1e-8)


def test_general_paths_without_the_lane_map():
    """A model outside the tree-row lane map (more than four trees, or a tree with more than 16 dofs) stores every
    constraint row in the compact chain form, solves with the level-parallel LDS routines and sweeps the rows
    serially.  The lane map is withheld from the two-agent level to run exactly those paths."""
    model, ora, emu = pair("two_agent.xml", lane_map=False)
    assert model.rowmap == 0
    rng = np.random.default_rng(21)
    for k in range(260):
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl; emu.ctrl[:] = ctrl
        img = emu.step(); ora.step()
        assert (img.nefc, img.ncon, img.niter) == (ora.nefc, ora.ncon, ora.niter), k
    assert ora.ncon > 0
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-9) and np.allclose(emu.qvel, ora.qvel, atol=1e-8)
    ctrl = rng.uniform(-1, 1, model.nu)
    ora.ctrl[:] = ctrl; emu.ctrl[:] = ctrl
    img = emu.step(dbg_stage=1); ora.step()
    n = ora.nefc
    assert n >= 4 and np.allclose(img.J()[:n], ora.efc_J[:n], atol=1e-12)


def _pose_with_many_rows_in_one_tree(model, packed):
    """An agent lowered onto the floor with slightly bent legs: four foot contacts (16 rows) plus its active joint
    limits put 17..32 rows in its tree.  Found with the oracle alone."""
    tree, gbody = model.body_treeid, model.geom_bodyid
    free = [j for j in range(model.njnt) if model.jnt_type[j] == 0]
    a0 = int(model.jnt_qposadr[free[0]])
    hinge_q = [int(model.jnt_qposadr[j]) for j in range(model.njnt) if model.jnt_type[j] == 3]
    rng = np.random.default_rng(1)
    probe = OracleEnv(packed)
    for dz in np.linspace(-0.55, -0.95, 9):
        for trial in range(6):
            q = model.qpos0.copy()
            q[a0 + 2] += dz
            if trial:
                q[hinge_q] += rng.uniform(-0.3, 0.3, len(hinge_q))
            probe.qpos[:] = q; probe.qvel[:] = 0
            probe.forward()
            rows = [0] * model.ntree
            for c in probe.contacts():
                t = max(int(tree[gbody[c["geom1"]]]), int(tree[gbody[c["geom2"]]]))
                if t >= 0:
                    rows[t] += 4
            if rows[0] == 16 and probe.nefc > sum(rows):
                return q
    return None


def test_wide_register_solver_for_17_to_32_rows_per_tree(emu_few):
    """More than 16 constraint rows in a tree (an ant on its four feet with joints at their limits): the solver then
    keeps 32 rows per tree in registers (pgs_wide_registers).  Sweep counts must still be the oracle's, step for
    step."""
    model, ora, emu = pair("two_agent.xml")
    q = _pose_with_many_rows_in_one_tree(model, blob.pack(model))
    assert q is not None
    ora.qpos[:] = q; emu.qpos[:] = q
    wide_steps = 0
    rng = np.random.default_rng(4)
    for k in range(60):
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl; emu.ctrl[:] = ctrl
        img = emu.step(); ora.step()
        assert (img.nefc, img.ncon, img.niter) == (ora.nefc, ora.ncon, ora.niter), k
        info_at = img._off("i_rowinfo")
        trees = (img.ints[info_at:info_at + img.nefc] >> 19) - 2
        per_tree = [int((trees == t).sum()) for t in range(model.ntree)]
        wide_steps += bool((trees >= 0).all() and 16 < max(per_tree) <= 32)
    assert wide_steps > 0
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-9) and np.allclose(emu.qvel, ora.qvel, atol=1e-8)


def test_coupling_rows_in_several_tree_pairs_at_once(emu_few):
    """Four agents, stacked in two pairs and one pair leaning on a third agent: rows that couple trees (0,1), (2,3)
    and (1,2) in the same step.  The trees sweep side by side with the coupling rows aligned in their lists
    (pgs_coupled_schedule); sweep counts and trajectories must match the oracle's plain serial sweep."""
    model, ora, emu = pair("four_agent.xml")
    free = [j for j in range(model.njnt) if model.jnt_type[j] == 0]
    assert len(free) == 4
    adr = [int(model.jnt_qposadr[j]) for j in free]
    q = model.qpos0.copy()
    base = q[adr[0]:adr[0] + 3].copy()
    q[adr[1]:adr[1] + 3] = base + np.array([0.15, 0.10, 0.55])          # agent 1 on agent 0
    q[adr[2]:adr[2] + 3] = base + np.array([1.05, 0.15, 0.00])          # agent 2 beside them, legs interleaved
    q[adr[3]:adr[3] + 3] = base + np.array([1.20, 0.25, 0.55])          # agent 3 on agent 2
    ora.qpos[:] = q; emu.qpos[:] = q
    pairs_seen = set()
    rng = np.random.default_rng(3)
    for k in range(45):
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl; emu.ctrl[:] = ctrl
        img = emu.step(); ora.step()
        assert (img.nefc, img.ncon, img.niter) == (ora.nefc, ora.ncon, ora.niter), k
        for c in ora.contacts():
            t1, t2 = (int(model.body_treeid[model.geom_bodyid[g]]) for g in (c["geom1"], c["geom2"]))
            if t1 >= 0 and t2 >= 0 and t1 != t2:
                pairs_seen.add((min(t1, t2), max(t1, t2)))
    assert len(pairs_seen) >= 2, pairs_seen
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-9) and np.allclose(emu.qvel, ora.qvel, atol=1e-8)


def test_more_than_sixteen_rows_per_tree_in_a_four_tree_model(emu_few):
    """17..32 rows in a tree of the 4-agent level: no idle lanes to spread a tree over, so a lane owns two rows of its
    tree (pgs_tall_registers, round 3; before: the sweep on the aligned schedule).  One agent is lowered onto the floor with bent legs (four
    foot contacts plus active joint limits) while the others fall; sweep counts and trajectories must be the
    oracle's."""
    model, ora, emu = pair("four_agent.xml")
    free = [j for j in range(model.njnt) if model.jnt_type[j] == 0]
    hinge = [j for j in range(model.njnt) if model.jnt_type[j] == 3]
    rng = np.random.default_rng(2)
    probe = OracleEnv(blob.pack(model))
    tree, gbody = model.body_treeid, model.geom_bodyid
    found = None
    probe.forward()
    limits_elsewhere = probe.nefc * (model.ntree - 1) // model.ntree      # the other agents stay as they are
    for dz in np.linspace(-0.72, -0.94, 12):
        for trial in range(30):
            q = model.qpos0.copy()
            q[int(model.jnt_qposadr[free[0]]) + 2] += dz
            if trial:
                hq = [int(model.jnt_qposadr[j]) for j in hinge if model.body_treeid[model.jnt_bodyid[j]] == 0]
                q[hq] += rng.uniform(-0.3, 0.3, len(hq))
            probe.qpos[:] = q; probe.qvel[:] = 0
            probe.forward()
            rows = [0] * model.ntree
            coupled = False
            for c in probe.contacts():
                t1, t2 = int(tree[gbody[c["geom1"]]]), int(tree[gbody[c["geom2"]]])
                coupled |= t1 >= 0 and t2 >= 0 and t1 != t2
                rows[max(t1, t2)] += 4
            in_tree_0 = rows[0] + probe.nefc - sum(rows) - limits_elsewhere
            if not coupled and probe.warnings == 0 and sum(rows[1:]) == 0 and 16 < in_tree_0 <= 32:
                found = q
                break
        if found is not None:
            break
    assert found is not None
    ora.qpos[:] = found; emu.qpos[:] = found
    many_row_steps = 0
    rng = np.random.default_rng(4)
    for k in range(50):
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl; emu.ctrl[:] = ctrl
        img = emu.step(); ora.step()
        assert (img.nefc, img.ncon, img.niter) == (ora.nefc, ora.ncon, ora.niter), k
        info_at = img._off("i_rowinfo")
        trees = (img.ints[info_at:info_at + img.nefc] >> 19) - 2
        per_tree = [int((trees == t).sum()) for t in range(model.ntree)]
        many_row_steps += bool((trees >= 0).all() and 16 < max(per_tree) <= 32)
    assert many_row_steps > 0
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-9) and np.allclose(emu.qvel, ora.qvel, atol=1e-8)


def test_in_launch_reset_equals_reset_then_step():
    """A copy flagged in the step's reset mask (mjrl_set_step_reset_mask) starts the launch from the reset image: the
    result is bit for bit that of reset (+ mj_forward) followed by the same step, whatever state the copy was in."""
    model, ora, emu = pair("two_agent.xml")
    reset_warm = emu.warm.copy()                     # what mj_forward leaves at the reset state
    rng = np.random.default_rng(3)
    scatter = np.array([[2, 3, 4, 5, 6, 7, 0, 1], [10, 11, 12, 13, 14, 15, 8, 9]], np.int32)
    for _ in range(40):                              # somewhere into an episode
        emu.step(actions=rng.uniform(-1, 1, (2, 8)), scatter=scatter, n_agent=2)
    assert emu.timestep[0] == 40
    actions = rng.uniform(-1, 1, (2, 8))
    emu.step(actions=actions, scatter=scatter, n_agent=2, reset_warm=reset_warm)
    _, _, fresh = pair("two_agent.xml")
    fresh.step(actions=actions, scatter=scatter, n_agent=2)
    assert emu.timestep[0] == 1
    for name in ("qpos", "qvel", "ctrl", "warm", "sens"):
        assert np.array_equal(getattr(emu, name), getattr(fresh, name)), name
    ora.ctrl[scatter.reshape(-1)] = actions.reshape(-1)
    ora.step()
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-12)


def test_target_and_pick_up_ops_follow_their_reference_dynamics():
    """OP_TARGET with an inventory (Testing/Pick_Up_Dynamic.py:15-41) plus a distance-decrease reward and a done on the
    agent's CURRENT target (target kind 2; Testing/SingleAgentTest.py:41-48), against a transcription of those plugins
    with the random choices drawn from the Python mix64 (dynamics.py) -- so this also pins the device generator."""
    from mjrl_amd import dynamics
    model, ora, emu = pair("two_agent.xml")
    names = model.names["body"]
    bodies = np.array([names.index("sender"), names.index("receiver")], np.int32)
    geoms = model.names["geom"]
    tags = [[(0, names.index("reference")), (0, names.index("choice_1")), (1, geoms.index("border5_geom")), (0, names.index("choice_2"))]]
    scatter = np.array([[2, 3, 4, 5, 6, 7, 0, 1], [10, 11, 12, 13, 14, 15, 8, 9]], np.int32)
    gather = np.full((2, 63), -1, np.int32)
    for a in range(2):
        gather[a, 0] = a
        gather[a, 1:31] = (1 << 24) | np.arange(30)
        gather[a, 31:59] = (2 << 24) | np.arange(28)
        gather[a, 59:63] = -2
    seed, thr, env_base = 12345, 9.0, 4096 + 17
    # slots: 0 current_target, 1 inventory, 2 distance
    prog_i = np.array([[4, 0, 0, 1, 0, 2, 0, 0], [2, 2, 0, 2, 1, 0, 0, 0], [3, 2, 0, 0, 0, 0, 0, 0]], np.int32)
    prog_f = np.array([[thr, 1.0, seed, 0], [2.0, 0, 0, 0], [2.5, 0, 0, 0]], np.float64)
    program = dict(prog_i=prog_i, prog_f=prog_f, n_slot=3, agent_body=bodies, agent_obs_len=np.array([59, 59], np.int32),
                   store=np.full((2, 3), np.nan), reward=np.zeros(2), term=np.zeros(2, np.uint8), trunc=np.zeros(2, np.uint8),
                   tags=tags, env_base=env_base)
    obs = np.zeros((2, 63))
    rng = np.random.default_rng(4)
    store = [dict(), dict()]

    def place(ref):
        kind, ident = ref
        return ora.xipos[ident] if kind == 0 else ora.geom_xpos[ident]

    toggles = 0
    for step in range(30):
        actions = rng.uniform(-1, 1, (2, 8))
        emu.step(actions=actions, scatter=scatter, n_agent=2, gather=gather, obs=obs, program=program)
        for a in range(2):
            ora.ctrl[scatter[a]] = actions[a]
        ora.step()
        rewards, dones = [0.0, 0.0], [False, False]
        for a in range(2):                                   # the dynamic, agent-minor
            st = store[a]
            if "current_target" not in st:
                st["current_target"] = int(dynamics.pick_of(dynamics.mix64(seed, env_base, a, step, 0), 4))
                st["inventory"] = 0.0
            here = ora.xipos[bodies[a]]
            dist = np.linalg.norm(here - place(tags[0][st["current_target"]]))
            if dist < thr:
                st["inventory"] = 1.0 - st["inventory"]
                rewards[a] += 1.0
                toggles += 1
                st["current_target"] = int(dynamics.pick_of(dynamics.mix64(seed, env_base, a, step, 1), 4))
                st["distance"] = np.linalg.norm(here - place(tags[0][st["current_target"]]))
            expect = np.concatenate([place(tags[0][st["current_target"]]), [st["inventory"]]])
            assert np.allclose(obs[a, 59:63], expect, atol=1e-12), (step, a)
        for a in range(2):                                   # the reward function
            dist = np.linalg.norm(ora.xipos[bodies[a]] - place(tags[0][store[a]["current_target"]]))
            if "distance" in store[a]:
                rewards[a] += 2.0 * (store[a]["distance"] - dist)
            store[a]["distance"] = dist
        for a in range(2):                                   # the done function
            dones[a] = np.linalg.norm(ora.xipos[bodies[a]] - place(tags[0][store[a]["current_target"]])) < 2.5
        assert np.allclose(program["reward"], rewards, atol=1e-10), step
        assert list(program["term"].astype(bool)) == dones
        for a in range(2):
            assert program["store"][a, 0] == store[a]["current_target"] and program["store"][a, 1] == store[a]["inventory"]
            assert np.isclose(program["store"][a, 2], store[a]["distance"], atol=1e-12)
    assert 4 <= toggles < 60


def test_longest_first_lookup_and_a_short_bucket_list():
    """The workgroup -> copy lookup of the longest-first dispatch: a workgroup finds its copy by walking the previous
    launch's work buckets from the heaviest down and taking the set bit of its rank in the bucket's row; when the bucket
    counts do not cover the workgroup (they always sum to n_env in launch_step -- this is the guard) the wave must leave
    without touching anything, not index with -1."""
    from tests.emu.emu import lib, _p
    model, ora, emu = pair("two_agent.xml")
    counts = np.zeros(16, np.int32)
    masks = np.zeros((16, 1), np.uint32)
    counts[7] = 1                                     # the one copy sits in bucket 7
    masks[7, 0] = 1
    lib().emu_set_lpt(_p(counts), _p(masks), 1)
    try:
        emu.step()
        ora.step()
        assert np.allclose(emu.qpos, ora.qpos, atol=1e-12) and emu.timestep[0] == 1
        counts[:] = 0                                 # tables that do not cover workgroup 0
        before = (emu.qpos.copy(), emu.qvel.copy(), emu.warm.copy(), emu.timestep.copy())
        emu.step()
        for was, now in zip(before, (emu.qpos, emu.qvel, emu.warm, emu.timestep)):
            assert np.array_equal(was, now)
    finally:
        lib().emu_set_lpt(None, None, 0)


def test_rank_of_a_copy_in_the_dispatch_bit_sets():
    """The lookup itself over many ranks: the device code's answer for (bucket counts, bit rows, workgroup id) equals
    the w-th copy of the buckets walked from the heaviest down, for rows longer than one block of 128 words, copies
    in the last word, empty buckets in between, and every workgroup id of a batch."""
    from tests.emu.emu import lib, _p
    rng = np.random.default_rng(5)
    for n_env in (1, 37, 4096, 5000):
        words = (n_env + 31) // 32
        bucket_of = rng.integers(0, 16, n_env)
        bucket_of[rng.random(n_env) < 0.5] = 3            # one crowded bucket
        counts = np.bincount(bucket_of, minlength=16).astype(np.int32)
        masks = np.zeros((16, words), np.uint32)
        for e, b in enumerate(bucket_of):
            masks[b, e >> 5] |= np.uint32(1 << (e & 31))
        order = [e for b in range(15, -1, -1) for e in range(n_env) if bucket_of[e] == b]
        ids = sorted(set([0, n_env - 1, n_env // 2] + list(rng.integers(0, n_env, 24))))
        got = np.zeros(len(ids), np.int32)
        lib().emu_lpt_lookup(_p(counts), _p(masks), words, n_env, _p(np.array(ids, np.int32)), len(ids), _p(got))
        assert list(got) == [order[w] for w in ids], n_env
    # a workgroup past the counts gets no copy
    got = np.zeros(1, np.int32)
    lib().emu_lpt_lookup(_p(np.zeros(16, np.int32)), _p(np.zeros((16, 1), np.uint32)), 1, 1, _p(np.array([0], np.int32)), 1, _p(got))
    assert got[0] == -1


def test_fixed_identity_frames_in_the_broad_phase_give_the_same_bits():
    """Pairs whose broad-phase test needs the frame of a geom that never rotates and has the identity orientation (the
    arena's floor and walls; bit 24 of the pair word) skip building and applying that rotation matrix.  With an exact
    identity the general form computes the same bits: a model with the flags cleared steps to the identical state."""
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    flagged = int(((model.pair_word >> 24) & 1).sum())
    assert flagged > 200                                        # every ant geom against the floor and the eight boxes
    plain = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    plain.arrays["pair_word"] = (plain.pair_word & ~np.int32(1 << 24)).astype(np.int32)
    a, b = EmuEnv(model, blob.pack(model)), EmuEnv(plain, blob.pack(plain))
    rng = np.random.default_rng(12)
    for env in (a, b):
        env.step(forward_only=True)
    most = 0
    for _ in range(260):
        ctrl = rng.uniform(-1, 1, model.nu)
        a.ctrl[:model.nu] = ctrl
        b.ctrl[:model.nu] = ctrl
        most = max(most, a.step().ncon)
        b.step()
    assert most > 0
    for name in ("qpos", "qvel", "warm", "sens"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_skipping_the_blocks_of_trees_out_of_reach_changes_nothing():
    """The bounding-sphere pairs between two kinematic trees are a block of the pair list that the broad phase skips when
    the trees' own bounding spheres are apart (mjcf._pair_layout).  A model whose blocks can never be skipped (reach =
    1e300) steps to identical bits -- with the two ants far apart (blocks skipped), while one walks into the other (live:
    contacts between the ants) and after they part again."""
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    assert model.ntp == 1 and 2.0 < model.tp_reach[0] < 3.0
    never = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    never.arrays["tp_reach"] = np.full(1, 1e300)
    a, b = EmuEnv(model, blob.pack(model)), EmuEnv(never, blob.pack(never))
    free = [j for j in range(model.njnt) if model.jnt_type[j] == mjcf.JNT_FREE]
    qa0, qa1 = int(model.jnt_qposadr[free[0]]), int(model.jnt_qposadr[free[1]])
    da1 = int(model.jnt_dofadr[free[1]])
    rng = np.random.default_rng(3)
    between = 0
    for env in (a, b):
        env.qpos[qa1:qa1 + 2] = env.qpos[qa0:qa0 + 2] + [1.6, 0.0]      # the second ant next to the first
        env.qpos[qa0 + 2] = env.qpos[qa1 + 2] = 0.6
        env.step(forward_only=True)
    tree_of = lambda g: int(model.body_treeid[model.geom_bodyid[g]])
    for k in range(300):
        ctrl = rng.uniform(-1, 1, model.nu)
        for env in (a, b):
            env.ctrl[:model.nu] = ctrl
            if 40 <= k < 120:
                env.qvel[da1] = -2.0                                        # pushed into the first ant, then let go
        img = a.step()
        b.step()
        between += sum(1 for g1, g2 in img.contact_geoms() if tree_of(g1) >= 0 and tree_of(g2) >= 0 and tree_of(g1) != tree_of(g2))
        for name in ("qpos", "qvel", "warm"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), (k, name)
    assert between > 10


def test_in_launch_reset_with_several_frames_per_step(emu_few):
    """skipFrames > 1: only the first launch of the step loads the reset image; the later frames continue from it, and
    the step counter, the data store and the fused program see a fresh episode."""
    model, ora, emu = pair("two_agent.xml")
    reset_warm = emu.warm.copy()
    rng = np.random.default_rng(9)
    scatter = np.array([[2, 3, 4, 5, 6, 7, 0, 1, -1], [10, 11, 12, 13, 14, 15, 8, 9, -1]], np.int32)
    gather = np.full((2, 60), -1, np.int32)
    for a in range(2):
        gather[a, 0] = a
        gather[a, 1:31] = (1 << 24) | np.arange(30)
        gather[a, 31:59] = (2 << 24) | np.arange(28)
        gather[a, 59] = -2
    bodies = np.array([model.names["body"].index("sender"), model.names["body"].index("receiver")], np.int32)

    def program(store):
        return dict(prog_i=np.array([[1, 8, 0, 0, 0, 0, 0, 0]], np.int32), prog_f=np.zeros((1, 4)), n_slot=1, agent_body=bodies,
                    agent_obs_len=np.array([59, 59], np.int32), store=store, reward=np.zeros(2), term=np.zeros(2, np.uint8),
                    trunc=np.zeros(2, np.uint8))
    store = np.full((2, 1), np.nan)
    obs = np.zeros((2, 60))
    for _ in range(25):
        emu.step(actions=np.concatenate([rng.uniform(-1, 1, (2, 8)), rng.uniform(0, 3, (2, 1))], axis=1), scatter=scatter,
                 n_agent=2, gather=gather, obs=obs, program=program(store), skip_frames=3)
    assert emu.timestep[0] == 25 and not np.isnan(store).any()
    actions = np.concatenate([rng.uniform(-1, 1, (2, 8)), [[2.4], [1.7]]], axis=1)
    store[:] = np.nan                                  # (the emulation's reset flag covers the state; the store row is the caller's)
    emu.step(actions=actions, scatter=scatter, n_agent=2, gather=gather, obs=obs, program=program(store), skip_frames=3,
             reset_warm=reset_warm)
    _, _, fresh = pair("two_agent.xml")
    store2, obs2 = np.full((2, 1), np.nan), np.zeros((2, 60))
    fresh.step(actions=actions, scatter=scatter, n_agent=2, gather=gather, obs=obs2, program=program(store2), skip_frames=3)
    assert emu.timestep[0] == 1
    for name in ("qpos", "qvel", "ctrl", "warm", "sens"):
        assert np.array_equal(getattr(emu, name), getattr(fresh, name)), name
    assert np.array_equal(obs, obs2) and np.array_equal(store, store2)
    assert obs[0, 59] == 0 and obs[1, 59] == 2         # a fresh episode: the sender has heard nothing yet


def test_reset_without_a_step_and_the_autoreset_kept_by_the_step(emu_few):
    """Round 3: a reset-mask byte of 2 resets a copy without stepping it (its outputs are the reset observation), and
    with mjrl_set_autoreset the step itself records that a copy's episode ended and resets it in the next step -- mode 1
    without a step (Gymnasium's next-step convention), mode 2 reset-then-step.  Device source on the CPU emulation
    against the oracle, one and two frames per step."""
    model = mjcf.compile_mjcf(levels.level_path("single_agent.xml"))
    packed = blob.pack(model)
    gather = np.array([[0] + [(1 << 24) | i for i in range(model.nq)] + [(2 << 24) | i for i in range(model.nv)]], np.int32)
    scatter = np.arange(model.nu, dtype=np.int32).reshape(1, -1)
    look = lambda o: np.concatenate([o.sensordata[:1], o.qpos, o.qvel])
    for frames in (1, 2):
        for mode in (1, 2):
            ora, emu = OracleEnv(packed), EmuEnv(model, packed)
            emu.step(forward_only=True)
            reset_warm, reset_sens = emu.warm.copy(), emu.sens.copy()
            first = look(ora)
            flag, episode = np.zeros(1, np.uint8), np.zeros(1, np.int32)
            obs = np.zeros((1, gather.shape[1]))
            program = None
            rng = np.random.default_rng(frames + 10 * mode)
            horizon, t = 3, 0
            for step in range(12):
                act = rng.uniform(-1, 1, (1, model.nu))
                pending = bool(flag[0])
                trunc = np.zeros((1,), np.uint8)
                prog = dict(prog_i=np.zeros((1, 8), np.int32), prog_f=np.zeros((1, 4)), n_op=0, n_slot=0, agent_body=np.zeros(1, np.int32),
                            agent_obs_len=np.array([gather.shape[1]], np.int32), store=np.zeros(1), reward=np.zeros(1),
                            term=np.zeros(1, np.uint8), trunc=trunc)
                emu.step(skip_frames=frames, actions=act, scatter=scatter, n_agent=1, gather=gather, obs=obs, max_steps=horizon,
                         program=prog, reset_warm=reset_warm, reset_sens=reset_sens, autoreset=(flag, mode, episode))
                if pending:
                    ora.reset()
                    t = 0
                    if mode == 1:                      # reset, not stepped: the reset observation, no flags, action ignored
                        assert np.allclose(obs[0], first, atol=1e-12) and not trunc[0] and not flag[0] and emu.timestep[0] == 0
                        continue
                ora.ctrl[:] = act[0]
                for _ in range(frames):
                    ora.step()
                t += 1
                assert np.allclose(obs[0], look(ora), atol=1e-9), (frames, mode, step)
                assert bool(trunc[0]) == (t == horizon + 1) and bool(flag[0]) == bool(trunc[0])
            assert episode[0] >= 2


SPRUNG = """<mujoco><option timestep="0.002"/>
<default><joint armature="0.05" damping="0.3" limited="true"/><geom density="60" friction="1 0.005 0.0001"/></default>
<worldbody><geom type="plane" size="5 5 0.1"/>
  <body name="hopper" pos="0 0 0.6"><freejoint/><geom type="sphere" size="0.12"/>
    <body pos="0.1 0 0"><joint type="hinge" axis="0 1 0" range="-60 60" stiffness="4" springref="25"/>
      <geom type="capsule" fromto="0 0 0 0.3 0 0" size="0.04"/>
      <body pos="0.3 0 0"><joint type="slide" axis="0 0 1" range="-0.2 0.2" stiffness="120" springref="-0.05"/>
        <geom type="capsule" fromto="0 0 0 0 0 -0.3" size="0.035"/></body></body>
    <body pos="-0.1 0 0"><joint type="hinge" axis="0 1 0" range="-60 60" stiffness="2.5" ref="10" springref="-20"/>
      <geom type="capsule" fromto="0 0 0 -0.3 0 0" size="0.04"/></body></body>
</worldbody></mujoco>"""


def test_joint_springs_in_the_device_source(emu_few):
    """Joint stiffness / springref (hinge and slide; a non-zero ref): the passive force of the smooth stage, device source
    against oracle through flight, landing and rest -- same counts every step, same trajectory."""
    model = mjcf.compile_mjcf_string(SPRUNG)
    assert np.count_nonzero(model.dof_stiffness) == 3 and model.qpos0[9] == pytest.approx(np.radians(10))
    packed = blob.pack(model)
    ora, emu = OracleEnv(packed), EmuEnv(model, packed)
    emu.step(forward_only=True)
    for step in range(700):
        img = emu.step()
        ora.step()
        assert (img.ncon, img.nefc, img.niter) == (ora.ncon, ora.nefc, ora.niter), step
    assert ora.ncon > 0
    assert np.abs(emu.qpos - ora.qpos).max() < 1e-9 and np.abs(emu.qvel - ora.qvel).max() < 1e-8


def test_one_agent_io_layout_in_the_device_source():
    """StepArgs::io_agent1 / obs_f32 (mjrl_set_io_layout) in the CPU emulation: with one agent's action row alone in the
    buffer (the other acting with 0) the step writes exactly that agent's observation row of the default layout -- the
    same doubles, or the same values as floats -- including the fused Language channel's slot."""
    from tests.emu import emu as emu_mod
    from tests.emu.batch import EmuBatch
    rng = np.random.default_rng(4)
    acts = rng.uniform(-1, 1, (25, 9)); acts[:, 8] = rng.integers(0, 3, 25) + 0.5
    for driven in (0, 1):
        for f32 in (0, 1):
            full = EmuBatch(levels.level_path("two_agent.xml"), ["sender", "receiver"], 1, language=True)
            lone = EmuBatch(levels.level_path("two_agent.xml"), ["sender", "receiver"], 1, language=True)
            n_obs = full.obs_dim
            for t in range(25):
                a_full = np.zeros((1, 2, 9)); a_full[0, driven] = acts[t]
                o_full, r = np.zeros((1, 2, n_obs)), np.zeros((1, 2))
                term, trunc = np.zeros((1, 2), np.uint8), np.zeros((1, 2), np.uint8)
                emu_mod.lib().emu_set_io_layout(0, 0)
                full.step_batched(a_full, o_full, r, term, trunc)
                # the one-agent layout: the action buffer's first act_dim doubles, the observation buffer's first obs_dim
                a_lone = np.zeros((1, 2, 9)); a_lone.reshape(-1)[:9] = acts[t]
                o_lone = np.zeros((1, 2, n_obs))
                emu_mod.lib().emu_set_io_layout(driven + 1, f32)
                try:
                    lone.step_batched(a_lone, o_lone, np.zeros((1, 2)), np.zeros((1, 2), np.uint8), np.zeros((1, 2), np.uint8))
                finally:
                    emu_mod.lib().emu_set_io_layout(0, 0)
                want = o_full[0, driven]
                if f32:
                    got = o_lone.reshape(-1).view(np.float32)[:n_obs]
                    assert np.array_equal(got, want.astype(np.float32)), (driven, t)
                else:
                    assert np.array_equal(o_lone.reshape(-1)[:n_obs], want), (driven, t)
                assert np.array_equal(lone.envs[0].qpos, full.envs[0].qpos)
