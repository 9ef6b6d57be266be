"""Differential fuzzing of the device step source (CPU lane emulation) against the oracle on small random scenes: one to
four free bodies carrying spheres, capsules and boxes, dropped next to each other onto a plane beside a static box, so that
every pair routine, rows inside one tree, rows that couple two, three or four trees, and the solver paths they select
(registers, wide, coupled schedule, LDS-resident fallback) all occur in combinations no hand-written level has.  Seeds are
fixed; a failure prints the scene's seed.  (Round 2's box tests found a solver bug this way: lanes of a tree the model does
not have walking tree 0's row list.)"""
import numpy as np
import pytest

from mjrl_amd import blob, mjcf
from oracle.oracle import OracleEnv
from tests.emu.emu import EmuEnv


def random_scene(rng, sensors=False, cameras=False):
    """``sensors``: every body also carries a site with one to three sensors of the kinds the levels use (rangefinder,
    touch, accelerometer, frame axes) -- drawn from a second generator.
    ``cameras``: every body carries a camera looking roughly at the middle of the scene, a fixed camera looks down on it,
    the geoms are coloured and one or two lights (spot or directional, with and without shadows, one of them possibly on a
    body) replace the default lighting -- a third generator."""
    n_body = int(rng.integers(1, 5))
    srng = np.random.default_rng(int(rng.integers(0, 2 ** 31))) if sensors else None
    crng = np.random.default_rng(int(rng.integers(0, 2 ** 31)) + 1) if cameras else None
    sensor_xml = []
    parts = []
    slots = rng.permutation(4)
    for b in range(n_body):
        # one body per quadrant, clear of its neighbours and of the floor at the start (no initial interpenetration: a
        # scene that starts with decimetres of overlap makes forces of 1e4 N, and there the solver's absolute 1e-10
        # cost guard acts on rounding noise -- oracle and kernel then stop at different sweeps, both "right")
        x, y = [(-0.3, -0.3), (0.3, -0.3), (-0.3, 0.3), (0.3, 0.3)][slots[b]] + rng.uniform(-0.04, 0.04, 2)
        z = rng.uniform(0.45, 0.7)
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        geoms = []
        for g in range(int(rng.integers(1, 3))):
            kind = rng.choice(["sphere", "capsule", "box"])
            off = rng.uniform(-0.08, 0.08, 3) if g else np.zeros(3)
            if kind == "sphere":
                geoms.append(f'<geom type="sphere" size="{rng.uniform(0.06, 0.12):.4f}" pos="{off[0]:.4f} {off[1]:.4f} {off[2]:.4f}"/>')
            elif kind == "capsule":
                a = rng.uniform(-0.12, 0.12, 3)
                geoms.append(f'<geom type="capsule" size="{rng.uniform(0.04, 0.07):.4f}" fromto="{off[0]:.4f} {off[1]:.4f} {off[2]:.4f} '
                             f'{off[0] + a[0]:.4f} {off[1] + a[1]:.4f} {off[2] + a[2] + 0.1:.4f}"/>')
            else:
                s = rng.uniform(0.05, 0.11, 3)
                e = rng.uniform(-40, 40, 3)
                geoms.append(f'<geom type="box" size="{s[0]:.4f} {s[1]:.4f} {s[2]:.4f}" pos="{off[0]:.4f} {off[1]:.4f} {off[2]:.4f}" '
                             f'euler="{e[0]:.2f} {e[1]:.2f} {e[2]:.2f}"/>')
        joint = '<joint type="free"/>' if rng.random() < 0.85 else \
            '<joint type="hinge" axis="0 1 0" damping="0.5" armature="0.1"/><joint type="slide" axis="0 0 1" damping="0.5"/>'
        site = ""
        if sensors:
            e = srng.uniform(-180, 180, 3)
            site = f'<site name="s{b}" pos="0 0 0" size="0.15" euler="{e[0]:.1f} {e[1]:.1f} {e[2]:.1f}"/>'
            for kind in srng.choice(["rangefinder", "touch", "accelerometer", "framexaxis", "frameyaxis", "framezaxis"],
                                    size=int(srng.integers(1, 4)), replace=False):
                if kind.startswith("frame"):
                    sensor_xml.append(f'<{kind} objtype="site" objname="s{b}"/>')
                else:
                    sensor_xml.append(f'<{kind} site="s{b}" cutoff="{srng.choice([0, 3, 50])}"/>')
        if cameras:
            geoms = [g.replace("<geom ", f'<geom rgba="{c[0]:.2f} {c[1]:.2f} {c[2]:.2f} 1" ') for g, c in zip(geoms, crng.uniform(0.1, 1, (len(geoms), 3)))]
            e = crng.uniform(-180, 180, 3)
            site += f'<camera name="cam{b}" pos="{crng.uniform(-0.1, 0.1):.3f} {crng.uniform(-0.1, 0.1):.3f} 0.15" euler="{e[0]:.1f} {e[1]:.1f} {e[2]:.1f}" fovy="{crng.uniform(30, 100):.1f}"/>'
            if b == 0 and crng.random() < 0.4:
                site += f'<light pos="0 0 0.3" dir="{crng.uniform(-1, 1):.2f} {crng.uniform(-1, 1):.2f} -0.5" diffuse=".6 .6 .6" cutoff="{crng.uniform(30, 80):.0f}"/>'
        parts.append(f'<body pos="{x:.4f} {y:.4f} {z:.4f}" quat="{q[0]:.5f} {q[1]:.5f} {q[2]:.5f} {q[3]:.5f}">{joint}{"".join(geoms)}{site}</body>')
    wall = '<body pos="0.75 0 0.3"><geom type="box" size="0.15 0.8 0.3"/></body>' if rng.random() < 0.7 else ""
    friction = rng.choice(["1 0.005 0.0001", "0.4 0.005 0.0001"])
    fixed = ""
    if cameras:
        wall = wall.replace("<geom ", '<geom rgba=".7 .6 .3 1" ')
        lx, ly = crng.uniform(-1, 1, 2)
        kind = 'directional="true"' if crng.random() < 0.3 else f'cutoff="{crng.uniform(25, 70):.0f}" exponent="{crng.choice([0, 5, 10])}"'
        shadow = 'castshadow="false"' if crng.random() < 0.2 else ""
        fixed = (f'<light pos="{lx:.2f} {ly:.2f} {crng.uniform(1.5, 3):.2f}" dir="{-lx * 0.3:.2f} {-ly * 0.3:.2f} -1" diffuse=".7 .7 .7" {kind} {shadow}/>'
                 f'<camera name="top" pos="{crng.uniform(-0.3, 0.3):.2f} {crng.uniform(-0.3, 0.3):.2f} 2.2" euler="0 0 {crng.uniform(-180, 180):.0f}" fovy="60"/>'
                 f'<camera name="side" pos="-1.6 {crng.uniform(-0.5, 0.5):.2f} 0.4" euler="90 -90 0" fovy="70"/>')
    return f"""
<mujoco><option timestep="0.002"/>
<default><geom density="300" margin="{rng.choice([0.0, 0.01])}" friction="{friction}"/></default>
<worldbody><geom type="plane" size="5 5 0.1"{' rgba=".4 .5 .4 1"' if cameras else ""}/>{fixed}{wall}{"".join(parts)}</worldbody>
{"<sensor>" + "".join(sensor_xml) + "</sensor>" if sensor_xml else ""}</mujoco>"""


def unexplained(got, ref, tol=2):
    """Pixels of `got` with a channel outside the range of the 3 x 3 pixels around them in `ref` (+- `tol` levels): not
    the one-pixel shift of an edge or of a steep gradient (a silhouette, a shadow's or a light cone's border, the floor's
    horizon, a small sphere's highlight) that fp32 rays against fp64 rays make."""
    H, W, _ = got.shape
    pad = np.pad(ref, ((1, 1), (1, 1), (0, 0)), mode="edge")
    stack = np.stack([pad[dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)])
    return ((got < stack.min(axis=0) - tol) | (got > stack.max(axis=0) + tol)).any(axis=-1)


def random_articulated_scene(rng):
    """One to three kinematic trees of two to seven bodies (free, hinge + slide or hinged roots; hinge and slide children up
    to two levels below the root, in random directions), with what the levels' ants have and what they do not: joint
    limits, damping, armature, springs with a rest angle, motors with gears and (mostly) clamped controls, limbs that
    reach their own tree's other limbs, the floor, a wall and the other trees.  Returns (xml, nu)."""
    n_tree = int(rng.integers(1, 4))
    slots = rng.permutation(4)
    motors = []
    count = [0]

    def limb(direction, length, depth, budget):
        """A child body at `direction` x 0.09 from its parent's frame: a capsule along `direction`, one or two joints."""
        k = count[0]
        count[0] += 1
        d = direction / np.linalg.norm(direction)
        joints = []
        for jn in range(int(rng.integers(1, 3)) if budget >= 2 else 1):
            name = f"j{k}_{jn}"
            if rng.random() < 0.8 or jn > 0:         # (two slides along one axis: a singular inertia matrix)
                axis = rng.normal(size=3) if rng.random() < 0.5 else np.eye(3)[int(rng.integers(0, 3))]
                lim = ""
                if rng.random() < 0.6:
                    lo, hi = -rng.uniform(5, 60), rng.uniform(5, 60)
                    lim = f' limited="true" range="{lo:.1f} {hi:.1f}"'
                spring = f' stiffness="{rng.uniform(1, 20):.2f}" springref="{rng.uniform(-20, 20):.1f}"' if rng.random() < 0.3 else ""
                joints.append(f'<joint name="{name}" type="hinge" axis="{axis[0]:.4f} {axis[1]:.4f} {axis[2]:.4f}" '
                              f'damping="{rng.uniform(0, 1):.3f}" armature="{rng.uniform(0, 0.05):.4f}"{lim}{spring}/>')
            else:
                joints.append(f'<joint name="{name}" type="slide" axis="{d[0]:.4f} {d[1]:.4f} {d[2]:.4f}" limited="true" '
                              f'range="-0.03 0.05" damping="{rng.uniform(0.5, 2):.3f}"/>')
            if rng.random() < 0.7:
                clamp = 'ctrllimited="true" ctrlrange="-1 1"' if rng.random() < 0.8 else 'ctrllimited="false"'
                motors.append(f'<motor joint="{name}" gear="{rng.uniform(2, 25):.2f}" {clamp}/>')
        end = d * length
        geom = f'<geom type="capsule" size="{rng.uniform(0.025, 0.04):.4f}" fromto="0 0 0 {end[0]:.4f} {end[1]:.4f} {end[2]:.4f}"/>'
        child = ""
        used = len(joints)
        if depth < 2 and budget - used >= 1 and rng.random() < 0.7:
            turn = d + rng.normal(size=3) * 0.5
            child = limb(turn, rng.uniform(0.08, 0.14), depth + 1, min(budget - used, 2))
            child = f'<body pos="{end[0]:.4f} {end[1]:.4f} {end[2]:.4f}">{child}</body>'
        return "".join(joints) + geom + child

    parts = []
    for t in range(n_tree):
        x, y = [(-0.35, -0.35), (0.35, -0.35), (-0.35, 0.35), (0.35, 0.35)][slots[t]] + rng.uniform(-0.03, 0.03, 2)
        q = rng.normal(size=4) * [1, 0.3, 0.3, 0.3]
        q /= np.linalg.norm(q)
        kind = rng.choice(["free", "free", "free", "planar", "hinged"])
        if kind == "free":
            z, root = rng.uniform(0.35, 0.5), '<joint type="free"/>'
        elif kind == "planar":
            z, root = rng.uniform(0.3, 0.4), ('<joint type="hinge" axis="0 1 0" damping="0.2" armature="0.05"/>'
                                               '<joint type="slide" axis="0 0 1" damping="0.5"/>')
        else:
            z, root = rng.uniform(0.3, 0.45), '<joint type="hinge" axis="1 0 0" damping="0.1"/>'
        torso = f'<geom type="sphere" size="{rng.uniform(0.06, 0.09):.4f}"/>' if rng.random() < 0.6 else \
            f'<geom type="box" size="{rng.uniform(0.05, 0.08):.4f} {rng.uniform(0.05, 0.08):.4f} {rng.uniform(0.03, 0.05):.4f}"/>'
        limbs = []
        # limbs leave the torso in separate directions (siblings are not parent and child: they may touch each other)
        dirs = [np.array(v, float) for v in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1))]
        for i in rng.permutation(5)[:int(rng.integers(1, 5))]:
            d = dirs[i] + rng.normal(size=3) * 0.15
            at = dirs[i] * 0.09
            limbs.append(f'<body pos="{at[0]:.4f} {at[1]:.4f} {at[2]:.4f}">{limb(d, rng.uniform(0.1, 0.16), 1, 2)}</body>')
        parts.append(f'<body pos="{x:.4f} {y:.4f} {z:.4f}" quat="{q[0]:.5f} {q[1]:.5f} {q[2]:.5f} {q[3]:.5f}">{root}{torso}{"".join(limbs)}</body>')
    wall = '<body pos="0.8 0 0.3"><geom type="box" size="0.15 0.8 0.3"/></body>' if rng.random() < 0.6 else ""
    friction = rng.choice(["1 0.005 0.0001", "0.5 0.005 0.0001"])
    xml = f"""
<mujoco><option timestep="0.002"/>
<default><geom density="400" margin="{rng.choice([0.0, 0.01])}" friction="{friction}"/></default>
<worldbody><geom type="plane" size="5 5 0.1"/>{wall}{"".join(parts)}</worldbody>
<actuator>{"".join(motors)}</actuator></mujoco>"""
    return xml, len(motors)


@pytest.mark.parametrize("block", range(6))
def test_random_scenes_step_like_the_oracle(block):
    paths = set()
    for k in range(8):
        seed = 1000 * block + k
        rng = np.random.default_rng(seed)
        model = mjcf.compile_mjcf_string(random_scene(rng), nconmax=24, njmax=120)
        packed = blob.pack(model)
        ora, emu = OracleEnv(packed), EmuEnv(model, packed)
        emu.step(forward_only=True)
        assert np.allclose(emu.warm, ora.qacc_warmstart, atol=1e-9), seed
        # push the bodies towards the middle so that they meet each other after landing
        for j in range(model.njnt):
            if model.jnt_type[j] == mjcf.JNT_FREE:
                qa, da = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
                for env in (ora, emu):
                    env.qvel[da:da + 2] = -2.0 * env.qpos[qa:qa + 2]
        worst = 0.0
        for step in range(260):
            img = emu.step()
            ora.step()
            assert (img.ncon, img.nefc) == (ora.ncon, ora.nefc), (seed, step)
            assert img.niter == ora.niter, (seed, step, img.niter, ora.niter)
            worst = max(worst, np.abs(emu.qpos - ora.qpos).max(), np.abs(emu.qvel - ora.qvel).max() * 1e-1)
            if ora.nefc:
                paths.add((model.ntree, model.rowmap, min(ora.nefc // 17, 3)))
        assert worst < 1e-8, (seed, worst)
        assert np.isfinite(emu.qpos).all()
    assert len(paths) >= 2


@pytest.mark.parametrize("block", range(2))
def test_random_scenes_with_sensors(block):
    """The same scenes with a site and one to three sensors per body (rangefinder, touch, accelerometer, frame axes, with
    and without cutoffs): the device source's sensor stage -- the sensors' records come as lane records -- against the
    oracle's readings, every step."""
    for k in range(6):
        seed = 9000 + 100 * block + k
        model = mjcf.compile_mjcf_string(random_scene(np.random.default_rng(seed), sensors=True), nconmax=24, njmax=120)
        assert model.nsensor >= 1
        packed = blob.pack(model)
        ora, emu = OracleEnv(packed), EmuEnv(model, packed)
        emu.step(forward_only=True)
        for j in range(model.njnt):
            if model.jnt_type[j] == mjcf.JNT_FREE:
                qa, da = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
                for env in (ora, emu):
                    env.qvel[da:da + 2] = -2.0 * env.qpos[qa:qa + 2]
        for step in range(220):
            img = emu.step()
            ora.step()
            assert (img.ncon, img.nefc, img.niter) == (ora.ncon, ora.nefc, ora.niter), (seed, step)
            assert np.allclose(emu.sens[:model.nsensordata], ora.sensordata, rtol=1e-8, atol=1e-8), (seed, step)


@pytest.mark.parametrize("block", range(3))
def test_random_articulated_scenes_step_like_the_oracle(block):
    """Random trees with joint limits, springs, damping, armature and motors under random controls (beyond their clamp):
    counts of every step and the states at the end, the device source against the oracle."""
    shapes = set()
    for k in range(4):
        seed = 3000 + 100 * block + k
        rng = np.random.default_rng(seed)
        xml, nu = random_articulated_scene(rng)
        model = mjcf.compile_mjcf_string(xml, nconmax=32, njmax=160)
        assert model.nu == nu and model.nv <= 64
        packed = blob.pack(model)
        ora, emu = OracleEnv(packed), EmuEnv(model, packed)
        emu.step(forward_only=True)              # (the reset's forward pass: the first step's warm start)
        crng = np.random.default_rng(seed + 1)
        worst = 0.0
        for step in range(220):
            if step % 10 == 0 and nu:
                ctrl = crng.uniform(-1.3, 1.3, nu)
                ora.ctrl[:nu] = ctrl
                emu.ctrl[:nu] = ctrl
            img = emu.step()
            ora.step()
            assert (img.ncon, img.nefc, img.niter) == (ora.ncon, ora.nefc, ora.niter), (seed, step)
            worst = max(worst, np.abs(emu.qpos - ora.qpos).max(), np.abs(emu.qvel - ora.qvel).max() * 1e-1)
        assert worst < 1e-8, (seed, worst)
        shapes.add((model.ntree, model.rowmap, model.nv))
    assert len(shapes) >= 2


def many_sensors_scene():
    """Two free bodies with twelve sites and 84 sensors between them -- more than a wavefront has lanes: sensors 64.. take the
    sensor stage's second pass, which reads their records from the model instead of the lane records."""
    rng = np.random.default_rng(77)
    bodies, sensors = [], []
    for b in range(2):
        sites = []
        for k in range(6):
            e = rng.uniform(-180, 180, 3)
            p = rng.uniform(-0.1, 0.1, 3)
            name = f"s{b}_{k}"
            sites.append(f'<site name="{name}" pos="{p[0]:.3f} {p[1]:.3f} {p[2]:.3f}" size="{rng.uniform(0.1, 0.25):.3f}" '
                         f'euler="{e[0]:.1f} {e[1]:.1f} {e[2]:.1f}"/>')
            for kind in ("rangefinder", "touch", "accelerometer", "framexaxis", "frameyaxis", "framezaxis", "rangefinder"):
                if kind.startswith("frame"):
                    sensors.append(f'<{kind} objtype="site" objname="{name}"/>')
                else:
                    sensors.append(f'<{kind} site="{name}" cutoff="{rng.choice([0, 2, 40])}"/>')
        geom = ('<geom type="box" size="0.12 0.09 0.07"/><geom type="sphere" size="0.08" pos="0.1 0.05 0.06"/>' if b == 0 else
                '<geom type="capsule" size="0.06" fromto="0 0 0 0.15 0.05 0.1"/>')
        bodies.append(f'<body pos="{0.25 * b - 0.1:.2f} 0.05 {0.5 + 0.3 * b:.2f}" euler="20 {30 + 50 * b} 10"><joint type="free"/>{geom}{"".join(sites)}</body>')
    return f"""
<mujoco><option timestep="0.002"/>
<default><geom density="300" friction="0.6 0.005 0.0001"/></default>
<worldbody><geom type="plane" size="5 5 0.1"/><body pos="0.6 0 0.3"><geom type="box" size="0.15 0.8 0.3"/></body>{"".join(bodies)}</worldbody>
<sensor>{"".join(sensors)}</sensor></mujoco>"""


def test_more_sensors_than_lanes():
    """84 sensors: the second pass of the sensor stage (records read from the model) against the oracle, every step."""
    model = mjcf.compile_mjcf_string(many_sensors_scene(), nconmax=24, njmax=120)
    assert model.nsensor == 84 and model.nsensordata == 2 * 6 * (1 + 1 + 3 + 9 + 1)
    packed = blob.pack(model)
    ora, emu = OracleEnv(packed), EmuEnv(model, packed)
    emu.step(forward_only=True)
    touched = 0.0
    for step in range(300):
        img = emu.step()
        ora.step()
        assert (img.ncon, img.nefc, img.niter) == (ora.ncon, ora.nefc, ora.niter), step
        assert np.allclose(emu.sens[:model.nsensordata], ora.sensordata, rtol=1e-8, atol=1e-8), step
        late = np.asarray(model.sensor_adr)[(np.asarray(model.sensor_type) == mjcf.SENS_TOUCH) & (np.arange(model.nsensor) >= 64)]
        touched = max(touched, float(ora.sensordata[late].max()))
    assert touched > 0            # a touch sensor of the second pass did read a force


TWO_MOTORS_ONE_JOINT = """
<mujoco><option timestep="0.002"/>
<worldbody><geom type="plane" size="5 5 0.1"/>
<body pos="0 0 0.6"><joint name="root" type="hinge" axis="0 1 0" damping="0.2"/><geom type="capsule" size="0.04" fromto="0 0 0 0.3 0 0"/>
  <body pos="0.3 0 0"><joint name="elbow" type="hinge" axis="0 1 0" damping="0.1" limited="true" range="-60 60"/>
    <geom type="capsule" size="0.03" fromto="0 0 0 0.25 0 0"/></body></body>
<body pos="0 0.5 0.3"><joint type="free"/><geom type="sphere" size="0.1"/></body></worldbody>
<actuator><motor joint="root" gear="3" ctrllimited="true" ctrlrange="-1 1"/><motor joint="elbow" gear="2"/>
<motor joint="root" gear="-1.5" ctrllimited="true" ctrlrange="-0.5 0.5"/></actuator></mujoco>"""


def test_two_motors_on_one_joint():
    """A dof driven by two motors (the stage that sums the actuator forces then scans the actuator list instead of reading
    the dof's one actuator from its lane record): their clamped, geared controls add up -- against the oracle, and against
    the closed form at the first step."""
    model = mjcf.compile_mjcf_string(TWO_MOTORS_ONE_JOINT)
    assert model.nu == 3 and list(model.arrays["dof_actid"][:2]) == [-2, 1]
    packed = blob.pack(model)
    ora, emu = OracleEnv(packed), EmuEnv(model, packed)
    emu.step(forward_only=True)
    rng = np.random.default_rng(3)
    for step in range(200):
        if step % 8 == 0:
            ctrl = rng.uniform(-1.5, 1.5, 3)
            ora.ctrl[:3] = ctrl
            emu.ctrl[:3] = ctrl
        img = emu.step()
        ora.step()
        assert (img.ncon, img.nefc, img.niter) == (ora.ncon, ora.nefc, ora.niter), step
        assert np.allclose(ora.qfrc_actuator[0], 3 * np.clip(ora.ctrl[0], -1, 1) - 1.5 * np.clip(ora.ctrl[2], -0.5, 0.5), atol=1e-14)
        assert np.abs(emu.qpos - ora.qpos).max() < 1e-9 and np.abs(emu.qvel - ora.qvel).max() < 1e-8, step
