"""Soak run: many thousand lockstep steps of a full batch with autoreset inside the launch (the bench's loop shape), checking
every few hundred steps that the state is finite, that no workgroup lost its copy, and reporting cap overflows and the
heaviest solves seen.  Usage: soak.py [level] [steps] [n_env] [autoreset: mask | kernel1 | kernel2]
(mask: the caller feeds the done flags back as the step-reset mask, round 2's loop; kernel1 / kernel2: the autoreset kept by
the step kernel itself, mjrl_set_autoreset modes 1 -- reset without a step -- and 2 -- reset, then step)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import levels
from mjrl_amd.mujoco_rl import MuJoCoRL
level = sys.argv[1] if len(sys.argv) > 1 else "two_agent"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
n_env = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
how = sys.argv[4] if len(sys.argv) > 4 else "mask"
agents = {"two_agent": ["sender", "receiver"], "four_agent": ["sender", "receiver", "agent_3", "agent_4"]}[level]
env = MuJoCoRL({"xmlPath": levels.level_path(level + ".xml"), "agents": agents, "numEnvs": n_env, "maxSteps": 1024})
env.reset()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
n_agent, act_dim = len(agents), env._handle.size("act_dim") if False else 8
done = torch.zeros(n_env, dtype=torch.uint8, device=dev)
if how == "mask":
    env._handle.set_step_reset_mask(done.data_ptr())
else:
    env._handle.set_autoreset(1 if how == "kernel1" else 2)
# staggered episodes: a different first length for every copy
env._handle.set_field("timestep", np.random.default_rng(0).integers(0, 1024, n_env).astype(np.int32))
out = None
worst = np.zeros(3, np.int64)
t0 = time.perf_counter()
for t in range(steps):
    a = torch.rand((n_env, n_agent, act_dim), dtype=torch.float64, device=dev, generator=g) * 2 - 1
    out = env.step_batched(a, *(out or ()))
    if how == "mask":
        torch.logical_or(out[2].any(1), out[3].any(1), out=done.view(torch.bool))
    if t % 500 == 499 or t == steps - 1:
        torch.cuda.synchronize()
        q = env._handle.get_field("qpos")
        stats = env._handle.get_field("solver_stats")
        assert np.isfinite(q).all() and np.isfinite(out[0].cpu().numpy()).all(), t
        worst = np.maximum(worst, stats[:, :3].max(0))
        over = env._handle.cap_overflows()
        ts = env._handle.get_field("timestep")
        assert ts.min() >= 0 and ts.max() <= 1025 and len(np.unique(ts)) > 100, (ts.min(), ts.max())      # episodes stay staggered and bounded
        print(f"step {t + 1}: finite, cap overflows {over}, max contacts / rows / sweeps so far {worst.tolist()}, "
              f"{(t + 1) * n_env / (time.perf_counter() - t0) / 1e6:.2f} M env-steps/s incl. action generation", flush=True)
print("soak ok")
