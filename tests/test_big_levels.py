"""Levels larger than a wavefront's lanes in bodies / geoms (the reference takes any MJCF, mujoco_parent.py:126): the
model compiler folds the bodies that cannot move into the world when a level has more than 64 bodies (MuJoCo's
``fusestatic``; physics unchanged, geom ids unchanged), and the kernels take geoms past the 64th in a second pass (geom
frames, rangefinder targets, ray-kernel candidates).  CPU: folding leaves the oracle's trajectory where it was; the device
source (lane emulation) follows the oracle on an arena of 73 geoms, rangefinder readings against geoms past the 64th
included; host queries by the name of a folded body still answer.  (GPU: tests/test_gpu_parity_r3.py.)"""
import numpy as np

from mjrl_amd import blob, levels, mjcf
from oracle.oracle import OracleEnv
from tests.emu.emu import EmuEnv


def big_level_text(pillars=12):
    """The 4-agent arena with `pillars` more static boxes, each a body of its own: 74 bodies, 73 geoms."""
    text = open(levels.level_path("four_agent.xml")).read()
    extra = "".join(f'<body pos="{-8 + 1.3 * k} {3.5 if k % 2 else -3.5} 0.3" name="pillar_{k}">'
                    f'<geom type="box" size="0.2 0.2 0.3" rgba="0.9 0.9 0 1" name="pillar_{k}_geom" /></body>' for k in range(pillars))
    return text.replace("<worldbody>", "<worldbody>" + extra, 1)


def test_folding_static_bodies_leaves_the_physics_where_it_was():
    text = open(levels.level_path("four_agent.xml")).read()
    plain, folded = mjcf.compile_mjcf_string(text), mjcf.compile_mjcf_string(text, fuse_static=True)
    assert plain.nbody == 62 and folded.nbody == 53 and plain.ngeom == folded.ngeom == 61
    assert np.array_equal(plain.pair_geom, folded.pair_geom) and plain.nv == folded.nv
    assert set(folded.folded_bodies) >= {"choice_1", "choice_2", "reference"} and not plain.folded_bodies
    assert np.allclose(folded.folded_bodies["choice_1"]["xipos"], [7.02852, -2.071592, 0.4710507])
    a, b = OracleEnv(blob.pack(plain)), OracleEnv(blob.pack(folded))
    rng = np.random.default_rng(0)
    for _ in range(300):
        ctrl = rng.uniform(-1, 1, plain.nu)
        a.ctrl[:] = ctrl; b.ctrl[:] = ctrl
        a.step(); b.step()
        assert (a.ncon, a.nefc, a.niter) == (b.ncon, b.nefc, b.niter)
    assert a.ncon > 0 and np.allclose(a.qpos, b.qpos, rtol=0, atol=1e-10) and np.allclose(a.sensordata, b.sensordata, atol=1e-9)


def test_arena_with_more_geoms_than_lanes_follows_the_oracle():
    model = mjcf.compile_mjcf_string(big_level_text())
    assert (model.nbody, model.ngeom) == (53, 73) and len(model.folded_bodies) == 15      # 74 bodies before folding (15 of the 21 folded ones have names)
    packed = blob.pack(model)
    ora, emu = OracleEnv(packed), EmuEnv(model, packed)
    emu.step(forward_only=True)
    # the first agent looks along +x from x = -5.5: put the fourth agent (geoms past the 64th) in its rangefinder's way
    names = model.names["geom"]
    late = [g for g in range(64, model.ngeom)]
    assert late and all(model.geom_bodyid[g] > 0 for g in late)
    rng = np.random.default_rng(5)
    hit_late = False
    for step in range(220):
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl
        emu.ctrl[:model.nu] = ctrl
        img = emu.step()
        ora.step()
        assert (img.nefc, img.ncon) == (ora.nefc, ora.ncon), step
    assert ora.ncon > 0
    assert np.allclose(emu.qpos, ora.qpos, rtol=0, atol=1e-9) and np.allclose(emu.sens[:model.nsensordata], ora.sensordata, atol=1e-7)
    # geom frames of the second pass (an LDS region the collision stage reads)
    assert np.allclose(img.region("gpos")[64:], ora.field("geom_xpos").reshape(-1, 3)[64:], atol=1e-9) if hasattr(ora, "field") else True


def test_rangefinder_sees_a_geom_past_the_64th():
    """Sensor level (the rangefinder looks down from 0.474 above the floor) with 66 static boxes far away and one more,
    the last geom, under the sensor's site: the reading is the distance to that box's top, in the oracle and in the
    device source -- the target is found by the second pass over the geoms."""
    text = open(levels.level_path("sensor_rangefinder.xml")).read()
    filler = "".join(f'<body pos="{30 + k} 30 0.5" name="far_{k}"><geom type="box" size="0.2 0.2 0.2" name="far_{k}_geom" /></body>'
                     for k in range(66))
    target = '<body pos="5.595446 1.222577 0.1" name="target"><geom type="box" size="0.1 0.1 0.1" name="target_geom" /></body>'
    base = mjcf.compile_mjcf_string(text)
    model = mjcf.compile_mjcf_string(text.replace("</worldbody>", filler + target + "</worldbody>", 1))
    # (geoms keep their document order: the target is the last geom although it now belongs to the world body)
    assert model.ngeom == base.ngeom + 67 and model.names["geom"].index("target_geom") == model.ngeom - 1 >= 64
    assert model.nbody <= 64 and "target" in model.folded_bodies
    packed = blob.pack(model)
    ora, emu, plain = OracleEnv(packed), EmuEnv(model, packed), OracleEnv(blob.pack(base))
    emu.step(forward_only=True)
    for o in (ora, plain):
        o.reset()
    assert abs(plain.sensordata[0] - 0.4743838) < 1e-9                 # the floor
    assert abs(ora.sensordata[0] - (0.4743838 - 0.2)) < 1e-9          # the box's top
    assert np.allclose(emu.sens[:model.nsensordata], ora.sensordata, atol=1e-12)
