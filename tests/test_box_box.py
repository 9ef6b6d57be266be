"""Box-box narrow phase (oracle/ora_collide.h ora_box_box_item; device csrc/mjrl_collide.h box_box_item).

The routine is this repo's own construction (separating axes, then the incident face clipped against the reference
face, or one edge-edge contact) -- MuJoCo's mjc_BoxBox cannot be consulted here, parity with it is UNPINNED -- so it is
pinned by known answers: a box resting on a box carries its weight, a pushed box stops at a wall, crossed boxes get the
corners of their overlap rectangle, tilted edges meet in one point; and the device routine must reproduce the oracle's
contacts bit for bit on random poses (through the CPU lane emulation here, on the GPU in tests/test_gpu_parity_r2.py)."""
import numpy as np
import pytest

from mjrl_amd import blob, mjcf
from oracle.oracle import OracleEnv
from tests.emu.emu import EmuEnv


def make(xml, **kw):
    model = mjcf.compile_mjcf_string(xml, **kw)
    packed = blob.pack(model)
    return model, packed, OracleEnv(packed)


def two_boxes(size1, size2, pos2, quat2=(1, 0, 0, 0), gravity="0 0 -9.81", margin=0.01, free1=False):
    joint1 = '<joint type="free" name="j1"/>' if free1 else ""
    return f"""
<mujoco><option timestep="0.002" gravity="{gravity}"/>
<default><geom margin="{margin}" density="1000" friction="1 0.005 0.0001"/></default>
<worldbody>
  <body name="lower" pos="0 0 0">{joint1}<geom type="box" size="{size1[0]} {size1[1]} {size1[2]}" name="lower_geom"/></body>
  <body name="upper" pos="{pos2[0]} {pos2[1]} {pos2[2]}" quat="{quat2[0]} {quat2[1]} {quat2[2]} {quat2[3]}">
    <joint type="free" name="j2"/><geom type="box" size="{size2[0]} {size2[1]} {size2[2]}" name="upper_geom"/></body>
</worldbody></mujoco>"""


def contact_points(env):
    return np.array(sorted([tuple(np.round(c["pos"], 9)) for c in env.contacts()]))


def test_small_box_on_large_box_rests_on_its_four_corners_and_carries_its_weight():
    model, packed, env = make(two_boxes((1, 1, 0.5), (0.2, 0.3, 0.1), (0.1, -0.2, 0.6 + 0.001)))
    assert blob._sizes(model)["pair_kmax"] == 16
    env.forward()
    assert env.ncon == 4
    pts = contact_points(env)
    assert np.allclose(sorted(pts[:, 0]), [-0.1, -0.1, 0.3, 0.3]) and np.allclose(sorted(pts[:, 1]), [-0.5, -0.5, 0.1, 0.1])
    for c in env.contacts():
        assert np.allclose(c["frame"][0], [0, 0, 1]) and c["dist"] == pytest.approx(0.001)
        assert (c["geom1"], c["geom2"]) == (0, 1)
    env.step(1500)
    weight = model.body_mass[2] * 9.81
    assert env.ncon == 4
    assert sum(c["normal_force"] for c in env.contacts()) == pytest.approx(weight, rel=1e-4)
    # (a contact with margin rests where dist - margin is slightly negative: the geoms float at the margin)
    assert np.abs(env.qvel).max() < 1e-4 and -0.005 < env.contacts()[0]["dist"] - 0.01 < 0


def test_large_box_on_small_box_gets_the_lower_boxes_corners():
    """The manifold comes from the reference face's corners when the incident face is the larger one."""
    model, packed, env = make(two_boxes((0.2, 0.3, 0.5), (1, 1, 0.1), (0, 0, 0.6 - 0.002)))
    env.forward()
    assert env.ncon == 4
    pts = contact_points(env)
    assert np.allclose(sorted(pts[:, 0]), [-0.2, -0.2, 0.2, 0.2]) and np.allclose(sorted(pts[:, 1]), [-0.3, -0.3, 0.3, 0.3])
    assert all(c["dist"] == pytest.approx(-0.002) and np.allclose(c["frame"][0], [0, 0, 1]) for c in env.contacts())
    env.step(1200)
    assert sum(c["normal_force"] for c in env.contacts()) == pytest.approx(model.body_mass[2] * 9.81, rel=1e-4)


def test_crossed_boxes_touch_on_the_overlap_rectangle():
    """Two bars crossed at right angles: no corner of either lies inside the other; the clipped manifold is the four
    corners of the overlap rectangle (edge / edge crossings)."""
    model, packed, env = make(two_boxes((1.0, 0.1, 0.1), (0.15, 0.8, 0.1), (0.2, 0, 0.2 - 0.001)))
    env.forward()
    assert env.ncon == 4
    pts = contact_points(env)
    assert np.allclose(sorted(pts[:, 0]), [0.05, 0.05, 0.35, 0.35]) and np.allclose(sorted(pts[:, 1]), [-0.1, -0.1, 0.1, 0.1])
    assert np.allclose(pts[:, 2], 0.1 - 0.0005)


def test_rotated_box_on_box_has_an_octagonal_manifold_inside_both_faces():
    c, s = np.cos(np.pi / 8), np.sin(np.pi / 8)          # 45 degrees about z
    model, packed, env = make(two_boxes((0.5, 0.5, 0.5), (0.5, 0.5, 0.1), (0, 0, 0.6 - 0.001), quat2=(c, 0, 0, s)), nconmax=12, njmax=48)
    env.forward()
    assert env.ncon == 8
    pts = contact_points(env)
    r = 0.5 * np.tan(np.pi / 8)
    expect = sorted([(0.5, r), (0.5, -r), (-0.5, r), (-0.5, -r), (r, 0.5), (-r, 0.5), (r, -0.5), (-r, -0.5)])
    assert np.allclose(sorted(map(tuple, np.round(pts[:, :2], 9))), expect, atol=1e-9)
    env.step(1500)
    assert sum(c["normal_force"] for c in env.contacts()) == pytest.approx(model.body_mass[2] * 9.81, rel=1e-4)


def test_edge_against_edge_is_one_contact_along_the_common_normal():
    """Upper box rolled 45 degrees about x, lower box rolled 45 degrees about y: the lowest edge of one crosses the top
    edge of the other; the contact sits between the two edges, normal along z."""
    c, s = np.cos(np.pi / 8), np.sin(np.pi / 8)
    xml = two_boxes((0.3, 0.3, 0.3), (0.3, 0.3, 0.3), (0, 0, 2 * 0.3 * np.sqrt(2) - 0.004), quat2=(c, s, 0, 0), gravity="0 0 0")
    xml = xml.replace('<body name="lower" pos="0 0 0">', f'<body name="lower" pos="0 0 0" quat="{c} 0 {s} 0">')
    model, packed, env = make(xml)
    env.forward()
    assert env.ncon == 1
    con = env.contacts()[0]
    assert np.allclose(con["frame"][0], [0, 0, 1], atol=1e-12) and con["dist"] == pytest.approx(-0.004)
    assert np.allclose(con["pos"], [0, 0, 0.3 * np.sqrt(2) - 0.002], atol=1e-12)


def test_pushed_box_stops_at_the_wall():
    """A free box driven along +x (the reference's freeJoint agents overwrite qvel every step, mujoco_parent.py:325)
    cannot enter a static wall: the wall's normal force cancels the approach velocity inside the step."""
    xml = """
<mujoco><option timestep="0.002"/>
<default><geom margin="0.01" density="5" friction="1 0.5 0.5"/></default>
<worldbody>
  <geom type="plane" size="10 10 0.1"/>
  <body pos="2 0 0.5"><geom type="box" size="0.25 5 0.5" name="wall"/></body>
  <body name="agent" pos="0.9 0.1 0.5"><joint type="free" name="root"/><geom type="box" size="0.5 0.5 0.5" name="agent_geom"/></body>
</worldbody></mujoco>"""
    model, packed, env = make(xml)
    far = []
    for _ in range(700):
        env.qvel[0] = 1.0                      # 1 m/s towards the wall, 0.35 m away
        env.step()
        far.append(env.qpos[0] + 0.5)          # the agent's +x face
    # the wall's face is at x = 1.75.  A body whose velocity is re-imposed every step is held where the soft contact's
    # reference acceleration cancels that velocity within one step: (b + k d) h = v with the default solref (0.02, 1):
    # b = 2 / (dmax tc) = 100, k = 1 / (dmax tc dr)^2 = 2500  ->  d = (1 / 0.002 - 100) / 2500 = 0.16 m into the wall
    # (the drive would carry it 0.4 m in 200 steps; against the wall it creeps by millimetres while it settles)
    assert abs(far[-1] - far[-200]) < 0.01 and far[-1] == pytest.approx(1.75 + 0.16, abs=0.03)
    pairs = {(c["geom1"], c["geom2"]) for c in env.contacts()}
    assert (1, 2) in pairs                     # wall against agent
    wall = [c for c in env.contacts() if (c["geom1"], c["geom2"]) == (1, 2)]
    assert len(wall) == 4 and all(np.allclose(c["frame"][0], [-1, 0, 0]) for c in wall)   # from the wall (geom1) to the agent
    for _ in range(500):                       # let go: the wall pushes the box back out
        env.step()
    assert env.qpos[0] + 0.5 < 1.75 + 0.011 and abs(env.qvel[0]) < 0.05
    # without the wall the same drive carries the box on
    model, packed, free = make(xml.replace('<body pos="2 0 0.5">', '<body pos="2 8 0.5">'))
    for _ in range(700):
        free.qvel[0] = 1.0
        free.step()
    assert free.qpos[0] + 0.5 > 2.7


def random_pose_xml(rng, n=1):
    a = rng.uniform(0.15, 0.6, 3)
    b = rng.uniform(0.15, 0.6, 3)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    q1 = rng.normal(size=4)
    q1 /= np.linalg.norm(q1)
    direction = rng.normal(size=3)
    direction /= np.linalg.norm(direction)
    gap = rng.uniform(0.4, 1.0) * (np.linalg.norm(a) + np.linalg.norm(b)) * rng.uniform(0.3, 1.0)
    p = direction * gap
    xml = two_boxes(a, b, p, quat2=q, gravity="0 0 0", margin=0.05, free1=True).replace('density="1000"', 'density="5"')
    return xml.replace('<body name="lower" pos="0 0 0">', f'<body name="lower" pos="0 0 0" quat="{q1[0]} {q1[1]} {q1[2]} {q1[3]}">')


def test_device_routine_reproduces_the_oracle_on_random_poses():
    """Contacts (count, order; dist, pos, frame to a few units in the last place: the two sources fuse products and
    sums by the same rule, but not every expression of the routine is shaped alike) from the device source in its CPU
    emulation; then a few steps of the resulting dynamics."""
    rng = np.random.default_rng(7)
    kinds = {"none": 0, "face": 0, "edge": 0}
    for trial in range(60):
        model, packed, ora = make(random_pose_xml(rng))
        emu = EmuEnv(model, packed)
        img = emu.step(forward_only=True)         # (the oracle ran its mj_forward when it was created)
        assert img.ncon == ora.ncon, trial
        cons = ora.contacts()
        dev = img.region("con")
        for k, c in enumerate(cons):
            assert np.isclose(dev[k, 0], c["dist"], rtol=1e-13, atol=1e-15) and np.allclose(dev[k, 1:4], c["pos"], rtol=1e-13, atol=1e-15), (trial, k)
            assert np.allclose(dev[k, 4:13].reshape(3, 3), c["frame"], rtol=1e-13, atol=1e-15), (trial, k)
        kinds["none" if not cons else ("edge" if len(cons) == 1 and abs(abs(cons[0]["frame"][0]).max() - 1) > 1e-3 else "face")] += 1
        if cons and trial % 6 == 0:
            for _ in range(5):
                img = emu.step()
                ora.step()
            assert np.allclose(emu.qpos, ora.qpos, atol=1e-10) and img.niter == ora.niter, trial
    assert kinds["face"] >= 10 and kinds["none"] >= 3, kinds


def test_free_box_agent_against_the_arena_walls():
    """The reference's sensor configurations drive a free BOX agent (Testing/sensor_test.py:20 on
    Testing/sensor_levels/Model1.xml): pushed at a border it must stop there, in the oracle and in the device source."""
    from mjrl_amd import levels
    model = mjcf.compile_mjcf(levels.level_path("sensor_touch.xml"))
    packed = blob.pack(model)
    ora, emu = OracleEnv(packed), EmuEnv(model, packed)
    emu.step(forward_only=True)
    assert blob._sizes(model)["pair_kmax"] == 16
    names = model.names["geom"]
    # drive towards border2 (y = +4.74, half width 0.25): the agent starts at y = 1.22, half size 0.5
    scatter = np.array([[0, 1, 5]], np.int32)
    emu.qpos[1] = ora.qpos[1] = 3.6              # 0.39 m short of the wall's face at y = 4.488
    for step in range(500):
        act = np.array([[0.0, 1.0, 0.0]])        # the action bound of a freeJoint agent (mujoco_parent.py:296)
        emu.step(actions=act, scatter=scatter, n_agent=1, scatter_mode=1)
        ora.qvel[[0, 1, 5]] = act[0]
        ora.step()
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-9)
    assert ora.qpos[1] + 0.5 < 4.738263 - 0.25 + 0.18           # held at the wall (0.16 m of soft-contact give, see above)
    hit = {(names[c["geom1"]], names[c["geom2"]]) for c in ora.contacts()}
    assert ("border2_geom", "receiver_geom") in hit
    before = ora.qpos[1]
    for step in range(200):
        emu.step(actions=act, scatter=scatter, n_agent=1, scatter_mode=1)
        ora.qvel[[0, 1, 5]] = act[0]
        ora.step()
    assert abs(ora.qpos[1] - before) < 0.01 and np.allclose(emu.qpos, ora.qpos, atol=1e-9)     # (0.4 m when free)
