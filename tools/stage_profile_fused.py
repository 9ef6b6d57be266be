"""Per-stage cycles of the step kernel in bench.py's default workload (two_agent + fused Language channel, actions from
the device), after 400 steps of random play.  Usage: stage_profile_fused.py [n_env]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ["MJRL_SPEC_FLAGS"] = (os.environ.get("MJRL_SPEC_FLAGS", "") + " -DMJRL_DIAG").strip()   # the diagnostic build of the specialised kernel
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import levels
from mjrl_amd.mujoco_rl import MuJoCoRL
from mjrl_amd.dynamics import Language
from mjrl_amd._capi import _host_ptr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = MuJoCoRL({"xmlPath": levels.level_path("two_agent.xml"), "agents": ["sender", "receiver"], "numEnvs": n,
                "skipFrames": 1, "maxSteps": 1024, "environmentDynamics": [Language]})
env.reset_batched()
dev = torch.device("cuda", 0)
h = env._handle
obs = torch.empty((n, 2, h.size("obs_dim")), dtype=torch.float64, device=dev)
rew = torch.empty((n, 2), dtype=torch.float64, device=dev)
term = torch.empty((n, 2), dtype=torch.uint8, device=dev)
trunc = torch.empty((n, 2), dtype=torch.uint8, device=dev)
rng = np.random.default_rng(0)
for t in range(400):
    env.step_batched(torch.from_numpy(rng.uniform(-1, 1, (n, 2, 9))).to(dev), obs, rew, term, trunc)
h.sync()
out, tot = np.zeros(len(h.STAGES), np.uint64), {}
for _ in range(5):
    a = torch.from_numpy(rng.uniform(-1, 1, (n, 2, 9))).to(dev)
    h._check(h._lib.mjrl_step_profile(h._h, ctypes.c_void_p(a.data_ptr()), 9, 1, _host_ptr(out), out.size))
    for k, v in zip(h.STAGES, out.tolist()):
        tot[k] = tot.get(k, 0) + v
print("kernel:", h.kernel, " lds doubles", h.size("lds_doubles"))
for k, v in tot.items():
    print(f"{k:12s} {v / 5 / n:10.0f} cycles/env-step  {100 * v / sum(tot.values()):5.1f} %")
print(f"total {sum(tot.values()) / 5 / n:.0f} cycles per env-step")
