"""The device step source under AddressSanitizer + UndefinedBehaviorSanitizer (CPU lane-emulation build, tests/emu
`make sanitized`): every LDS access of a copy lands inside the copy's image -- the image is a hand-managed union of
lifetimes (mjrl_step.h `Lay`), an index past its end is a heap overflow here and a silent neighbour-corrupting access on
the GPU -- and no signed overflow / bad shift / misaligned access is executed.  GPU sanitizers are not available on the
pool, so this is where they run.  The sanitized library needs libasan loaded first, hence the child process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import blob, levels, mjcf
from tests.emu.emu import EmuEnv
from tests.emu.batch import EmuBatch

from tests.emu import emu as emu_mod
rng = np.random.default_rng(0)
# short trajectories from just above the floor, so that contacts, limit rows, coupling rows and the cap paths are reached
# within a few dozen steps (a lane switch costs a system call here; the unsanitized suite runs the long trajectories);
# the last two with the solver forms of the build for batches of at most one wave per SIMD (StepArgs::few)
for level, kw, steps, few in (("two_agent.xml", {}, 240, 0), ("four_agent.xml", {}, 160, 0),
                              ("sensor_touch.xml", dict(nconmax=2, njmax=8), 60, 0), ("two_agent_3sensors.xml", {}, 120, 0),
                              ("single_agent.xml", dict(lane_map=False), 120, 0), ("two_agent.xml", {}, 240, 1),
                              ("four_agent.xml", {}, 160, 1)):
    emu_mod.lib().emu_set_few(few)
    model = mjcf.compile_mjcf(levels.level_path(level), **kw)
    env = EmuEnv(model, blob.pack(model))
    for j in range(model.njnt):
        if model.jnt_type[j] == mjcf.JNT_FREE and not level.startswith("sensor_"):
            env.qpos[model.jnt_qposadr[j] + 2] = 0.14
    env.step(forward_only=True)
    most = 0
    for t in range(steps):
        env.ctrl[:model.nu] = rng.uniform(-1, 1, model.nu)
        most = max(most, env.step().ncon)
    assert np.isfinite(env.qpos).all() and most > 0, (level, most)
emu_mod.lib().emu_set_few(0)
# random scenes (tests/test_fuzz_scenes.py): box-box items, rows that couple two to four trees, hinge / slide joints
from tests.test_fuzz_scenes import random_scene
for seed in (3, 1002, 3001, 5004):
    model = mjcf.compile_mjcf_string(random_scene(np.random.default_rng(seed)), nconmax=24, njmax=120)
    env = EmuEnv(model, blob.pack(model))
    for j in range(model.njnt):
        if model.jnt_type[j] == mjcf.JNT_FREE:
            qa, da = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
            env.qvel[da:da + 2] = -2.0 * env.qpos[qa:qa + 2]
    env.step(forward_only=True)
    for t in range(200):
        env.step()
    assert np.isfinite(env.qpos).all(), seed
# the Runge-Kutta level
model = mjcf.compile_mjcf(levels.level_path("ant.xml"))
env = EmuEnv(model, blob.pack(model))
env.step(forward_only=True)
for t in range(60):
    env.ctrl[:model.nu] = rng.uniform(-1, 1, model.nu)
    env.step()
assert np.isfinite(env.qpos).all()
# the fused program, the gather / scatter tables and the in-launch reset
batch = EmuBatch(levels.level_path("two_agent.xml"), ["sender", "receiver"], 2, language=True)
obs, rew = np.zeros((2, 2, batch.obs_dim)), np.zeros((2, 2))
term, trunc = np.zeros((2, 2), np.uint8), np.zeros((2, 2), np.uint8)
for t in range(6):
    batch.set_step_reset_mask(np.array([t == 3, t == 4], np.uint8))
    batch.step_batched(rng.uniform(-1, 1, (2, 2, 9)), obs, rew, term, trunc)
# the one-agent I/O layout (StepArgs::io_agent1 / obs_f32): one agent's action row in, its observation row out, as floats
for agent1, f32 in ((1, 0), (2, 1)):
    emu_mod.lib().emu_set_io_layout(agent1, f32)
    for t in range(4):
        batch.step_batched(rng.uniform(-1, 1, (2, 2, 9)), obs, rew, term, trunc)
emu_mod.lib().emu_set_io_layout(0, 0)
# a reset without a step (mask byte 2) and the autoreset kept by the step itself, two frames per step
model = mjcf.compile_mjcf(levels.level_path("single_agent.xml"))
env = EmuEnv(model, blob.pack(model))
env.step(forward_only=True)
warm0, sens0 = env.warm.copy(), env.sens.copy()
gather = np.array([[0] + [(1 << 24) | i for i in range(model.nq)] + [(2 << 24) | i for i in range(model.nv)]], np.int32)
scatter = np.arange(model.nu, dtype=np.int32).reshape(1, -1)
flag, episode, row = np.zeros(1, np.uint8), np.zeros(1, np.int32), np.zeros((1, gather.shape[1]))
for mode in (1, 2):
    for t in range(9):
        prog = dict(prog_i=np.zeros((1, 8), np.int32), prog_f=np.zeros((1, 4)), n_op=0, n_slot=0, agent_body=np.zeros(1, np.int32),
                    agent_obs_len=np.array([gather.shape[1]], np.int32), store=np.zeros(1), reward=np.zeros(1),
                    term=np.zeros(1, np.uint8), trunc=np.zeros(1, np.uint8))
        env.step(skip_frames=2, actions=rng.uniform(-1, 1, (1, model.nu)), scatter=scatter, n_agent=1, gather=gather, obs=row,
                 max_steps=2, program=prog, reset_warm=warm0, reset_sens=sens0, autoreset=(flag, mode, episode))
    env.step(actions=rng.uniform(-1, 1, (1, model.nu)), scatter=scatter, n_agent=1, gather=gather, obs=row, reset_warm=warm0,
             reset_kind=2, reset_sens=sens0)
assert episode[0] >= 4 and np.isfinite(row).all()
# an arena with more geoms than lanes (second pass over geoms 64..72; static bodies folded into the world)
from tests.test_big_levels import big_level_text
model = mjcf.compile_mjcf_string(big_level_text())
env = EmuEnv(model, blob.pack(model))
for j in range(model.njnt):
    if model.jnt_type[j] == mjcf.JNT_FREE:
        env.qpos[model.jnt_qposadr[j] + 2] = 0.14
env.step(forward_only=True)
most = 0
for t in range(60):
    env.ctrl[:model.nu] = rng.uniform(-1, 1, model.nu)
    img = env.step()
    most = max(most, img.nefc)
assert np.isfinite(env.qpos).all() and most > 16
print("sanitized run ok")
"""


def test_device_source_is_clean_under_asan_and_ubsan():
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan is not installed")
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emu"), "-s", "sanitized"], check=True)
    env = dict(os.environ, LD_PRELOAD=libasan, MJRL_EMU_SANITIZED="1",
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0:exitcode=23",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=24")
    res = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0 and "sanitized run ok" in res.stdout, (res.stdout + res.stderr)[-4000:]
