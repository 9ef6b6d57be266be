"""Rate of the vector-env adapter (wrappers.BatchedVectorEnv) at 4096 copies: numpy in / numpy out over the pinned host
buffers, and torch in / torch out in HBM; next-step autoreset kept by the kernel.  Usage: vec_env_rate.py [level] [n_env]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import levels
from mjrl_amd.mujoco_rl import MuJoCoRL
from mjrl_amd.wrappers import BatchedVectorEnv

level = sys.argv[1] if len(sys.argv) > 1 else "two_agent.xml"
n_env = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
agents = {"two_agent.xml": ["sender", "receiver"], "single_agent.xml": ["sender"]}[level]
for path, copy, dtype in (("numpy", True, np.float64), ("numpy", False, np.float64), ("numpy", True, np.float32),
                          ("numpy", False, np.float32), ("torch", True, np.float64), ("torch", False, np.float64),
                          ("torch", True, np.float32)):
    vec = BatchedVectorEnv(MuJoCoRL({"xmlPath": levels.level_path(level), "agents": agents, "numEnvs": n_env, "maxSteps": 1024}),
                           agent="sender", copy=copy, obs_dtype=dtype)
    vec.reset()
    rng = np.random.default_rng(0)
    ring = rng.uniform(-1, 1, (64, n_env, 8))
    if path == "torch":
        ring = torch.from_numpy(ring).cuda()
    # spread the copies over their episodes so that the resets do not come all at once
    vec.environment._handle.set_field("timestep", (np.arange(n_env) * 1024 // n_env).astype(np.int32))
    for i in range(1100):
        vec.step(ring[i % 64])
    torch.cuda.synchronize()
    steps = 1000
    t0 = time.perf_counter()
    total = 0.0
    for i in range(steps):
        obs, rew, term, trunc, info = vec.step(ring[i % 64])
    if path == "torch":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{path:6s} copy={copy!s:5s} {level} {n_env} copies: {n_env * steps / dt / 1e6:.2f} M env-steps/s ({dt / steps * 1e3:.3f} ms per step), "
          f"obs {tuple(obs.shape)} {obs.dtype}")
    vec.close()
