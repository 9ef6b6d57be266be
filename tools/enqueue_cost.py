"""Host time to enqueue one bench step (the loop body of bench.py's timed region) against the GPU time of the step: how much
slack the launching thread has.  Usage: enqueue_cost.py [level] [n_env]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import levels
from mjrl_amd.mujoco_rl import MuJoCoRL
from mjrl_amd.dynamics import Language
level = sys.argv[1] if len(sys.argv) > 1 else "two_agent"
n_env = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
agents = {"two_agent": ["sender", "receiver"], "four_agent": ["sender", "receiver", "agent_3", "agent_4"]}[level]
env = MuJoCoRL({"xmlPath": levels.level_path(level + ".xml"), "agents": agents, "numEnvs": n_env, "maxSteps": 1024,
                "environmentDynamics": [Language]})
env.reset()
dev = torch.device("cuda", 0)
act = torch.rand((16, n_env, len(agents), 9), dtype=torch.float64, device=dev) * 2 - 1
mask = torch.zeros((16, n_env), dtype=torch.uint8, device=dev)
out = None
stream = torch.cuda.current_stream(dev)
def body(i):
    global out
    env._handle.set_step_reset_mask(mask[i % 16].data_ptr())
    with torch.cuda.stream(stream):
        out = env.step_batched(act[i % 16], *(out or ()))
for i in range(50):
    body(i)
torch.cuda.synchronize()
for n in (20, 200):
    t0 = time.perf_counter()
    for i in range(n):
        body(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: enqueue {1e6 * (t1 - t0) / n:.1f} us per step on the host, {1e6 * (t2 - t0) / n:.1f} us per step until the GPU is done")
