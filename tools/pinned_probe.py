"""Where the time of the pinned host path goes: mjrl_step_pinned alone, the numpy copy of the actions alone, and the
device-resident path at the same point of the episode."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import levels
from mjrl_amd.mujoco_rl import MuJoCoRL
n_env = 4096
env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": ["sender", "receiver"], "numEnvs": n_env})
env.reset()
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, (32, n_env, 2, 8))
h = env._handle
bufs = h.host_buffers(8)
for i in range(300):
    env.step_batched(acts[i % 32])
K = 300
t = time.perf_counter()
for i in range(K):
    h.step_pinned(8, 1)                 # no action copy, no python wrapper: the C call alone
dt = time.perf_counter() - t
print(f"step_pinned alone (actions already in the pinned buffer): {dt / K * 1e3:.3f} ms")
t = time.perf_counter()
for i in range(K):
    np.copyto(bufs[0], acts[i % 32])
dt = time.perf_counter() - t
print(f"np.copyto of the actions alone: {dt / K * 1e3:.3f} ms")
dacts = torch.from_numpy(acts).cuda()
out = env.step_batched(dacts[0])
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(K):
    out = env.step_batched(dacts[i % 32], *out)
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"device path: {dt / K * 1e3:.3f} ms")
