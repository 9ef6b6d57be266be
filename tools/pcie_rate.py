"""PCIe-inclusive rates of the two host entries: mjrl_step_host (numpy arrays of the caller: pageable copies in and out)
and mjrl_step_pinned (the handle's pinned buffers, read and written by the kernel itself; numpy views out)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import levels
from mjrl_amd.mujoco_rl import MuJoCoRL
n_env = 4096
env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": ["sender", "receiver"], "numEnvs": n_env})
env.reset()
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, (32, n_env, 2, 8))
obs = np.zeros((n_env, 2, 59)); rew = np.zeros((n_env, 2)); term = np.zeros((n_env, 2), np.uint8); trunc = np.zeros((n_env, 2), np.uint8)
for i in range(20):
    env.step_batched(acts[i % 32], obs, rew, term, trunc)
t = time.perf_counter()
K = 300
for i in range(K):
    env.step_batched(acts[i % 32], obs, rew, term, trunc)
dt = time.perf_counter() - t
print(f"host-buffer path: {dt / K * 1e3:.3f} ms/step, {n_env * K / dt:.3e} env-steps/s (PCIe copies of 0.5 MB in, 3.9 MB out per step included)")
for i in range(20):
    env.step_batched(acts[i % 32])
t = time.perf_counter()
for i in range(K):
    out = env.step_batched(acts[i % 32])
dt = time.perf_counter() - t
print(f"pinned-buffer path: {dt / K * 1e3:.3f} ms/step, {n_env * K / dt:.3e} env-steps/s (actions copied into the pinned buffer by numpy, "
      f"results read in place; obs checksum {float(out[0].sum()):.6f})")
