"""The Runge-Kutta integrator (<option integrator="RK4">: benchmarking/levels/Ant.xml:3, dt 0.01) in the oracle and in
the device source.  Known answers pin the oracle: RK4 integrates a constant acceleration exactly (free fall follows
z0 - g t^2 / 2, where the semi-implicit Euler step follows the discrete sum), and it keeps the energy of a frictionless
pendulum orders of magnitude better than Euler does at the same step.  The device source (CPU lane emulation here, the
GPU in tests/test_gpu_parity_r2.py) must follow the oracle on the ant level, contacts included."""
import numpy as np
import pytest

from mjrl_amd import blob, levels, mjcf
from oracle.oracle import OracleEnv
from tests.emu.emu import EmuEnv

BALL = """
<mujoco><option timestep="0.01" integrator="{integ}"/><worldbody>
  <body name="ball" pos="0 0 5"><joint type="free" name="root"/><geom type="sphere" size="0.1" density="1000"/></body>
</worldbody></mujoco>"""

PENDULUM = """
<mujoco><option timestep="0.01" integrator="{integ}"/><worldbody>
  <body name="arm" pos="0 0 2"><joint type="hinge" name="pivot" axis="0 1 0" damping="0" armature="0" limited="false"/>
    <geom type="capsule" fromto="0 0 0 0 0 -1" size="0.05" density="1000" contype="0" conaffinity="0"/></body>
</worldbody></mujoco>"""


def make(xml):
    model = mjcf.compile_mjcf_string(xml)
    return model, OracleEnv(blob.pack(model))


def test_rk4_free_fall_is_exact_and_the_sensor_level_defaults_to_euler():
    model, env = make(BALL.format(integ="RK4"))
    assert model.integrator == mjcf.INT_RK4
    n, h, g = 120, 0.01, 9.81
    env.step(n)
    t = n * h
    assert env.qpos[2] == pytest.approx(5.0 - 0.5 * g * t * t, rel=1e-13)
    assert env.qvel[2] == pytest.approx(-g * t, rel=1e-13) and env.time == pytest.approx(t)
    _, euler = make(BALL.format(integ="Euler"))
    euler.step(n)
    assert euler.qpos[2] == pytest.approx(5.0 - g * h * h * n * (n + 1) / 2, rel=1e-13)      # the discrete sum, not t^2 / 2
    assert mjcf.compile_mjcf(levels.level_path("two_agent.xml")).integrator == mjcf.INT_EULER
    assert mjcf.compile_mjcf(levels.level_path("ant.xml")).integrator == mjcf.INT_RK4


def test_rk4_keeps_the_pendulum_energy_far_better_than_euler():
    drift = {}
    for integ in ("RK4", "Euler"):
        model, env = make(PENDULUM.format(integ=integ))
        env.qpos[0] = 1.0

        def energy():
            env.forward()
            height = env.xipos[1][2]
            return 0.5 * env.qvel @ env.qMdense @ env.qvel + model.body_mass[1] * 9.81 * height

        e0 = energy()
        env.step(400)                       # ~2 periods
        drift[integ] = abs(energy() - e0) / abs(e0)
    assert drift["RK4"] < 1e-7 and drift["Euler"] > 100 * drift["RK4"]


def test_device_source_follows_the_oracle_on_the_ant_level():
    model = mjcf.compile_mjcf(levels.level_path("ant.xml"))
    packed = blob.pack(model)
    ora, emu = OracleEnv(packed), EmuEnv(model, packed)
    emu.step(forward_only=True)
    assert np.allclose(emu.warm, ora.qacc_warmstart, atol=1e-10)
    rng = np.random.default_rng(5)
    most = 0
    for step in range(120):                 # 1.2 s: the ant drops from 0.75 m, lands and struggles
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl
        emu.ctrl[:model.nu] = ctrl
        img = emu.step()
        ora.step()
        most = max(most, ora.ncon)
        if step % 20 == 19:
            assert np.allclose(emu.qpos, ora.qpos, atol=1e-9) and np.allclose(emu.qvel, ora.qvel, atol=1e-8), step
            assert img.ncon == ora.ncon and img.niter == ora.niter
    assert most > 0 and emu.timestep[0] == 120
    # skipFrames: two frames per step() call = two Runge-Kutta steps
    emu.step(skip_frames=2)
    ora.step(2)
    assert np.allclose(emu.qpos, ora.qpos, atol=1e-9) and emu.timestep[0] == 121


def test_sensors_and_observations_come_from_the_first_pass():
    """A Runge-Kutta frame runs four forward passes; sensordata is the first one's (mj_step's own mj_forward), the
    observation gather of the last launch must read those values."""
    text = open(levels.level_path("two_agent_3sensors.xml")).read().replace("<default>", '<option integrator="RK4"/>\n  <default>', 1)
    model = mjcf.compile_mjcf_string(text)
    assert model.integrator == mjcf.INT_RK4 and model.nsensordata == 10
    packed = blob.pack(model)
    ora, emu = OracleEnv(packed), EmuEnv(model, packed)
    emu.step(forward_only=True)
    gather = np.full((2, 5 + 30 + 28), -1, np.int32)
    for a, idx in enumerate(([0, 2, 4, 5, 6], [1, 3, 7, 8, 9])):
        gather[a, :5] = idx
        gather[a, 5:35] = (1 << 24) | np.arange(30)
        gather[a, 35:63] = (2 << 24) | np.arange(28)
    obs = np.zeros((2, 63))
    rng = np.random.default_rng(1)
    for step in range(260):
        ctrl = rng.uniform(-1, 1, model.nu)
        ora.ctrl[:] = ctrl
        emu.ctrl[:model.nu] = ctrl
        emu.step(gather=gather, obs=obs, n_agent=2)
        ora.step()
        if step % 52 == 51:
            assert np.allclose(emu.sens[:10], ora.sensordata, atol=1e-7)
            for a, idx in enumerate(([0, 2, 4, 5, 6], [1, 3, 7, 8, 9])):
                assert np.allclose(obs[a], np.concatenate([ora.sensordata[idx], ora.qpos, ora.qvel]), atol=1e-7)
    assert ora.ncon > 0 and np.abs(ora.sensordata[4:10]).max() > 0       # the accelerometers read something
