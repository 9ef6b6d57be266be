"""Bit-level regression aid: step a few levels on the CPU lane emulation of the device source with a fixed action
stream and save every step's state, sensors and solver counts.  Run in two trees (before / after a change that must
not alter results) and compare the files: tools/emu_trajectory_dump.py out.npz [steps]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import blob, levels, mjcf
from tests.emu.emu import EmuEnv

out = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
res = {}
for name in ["two_agent.xml", "four_agent.xml", "sensor_touch.xml", "sensor_accelerometer.xml", "sensor_rangefinder.xml",
             "sensor_framexaxis.xml", "two_agent_3sensors.xml", "single_agent.xml"]:
    model = mjcf.compile_mjcf(levels.level_path(name))
    env = EmuEnv(model, blob.pack(model))
    rng = np.random.default_rng(7)
    # start low so that contacts, limits and the solver paths are exercised within the run
    rec = []
    for k in range(steps):
        env.ctrl[:] = rng.uniform(-1, 1, env.ctrl.shape)
        img = env.step()
        rec.append(np.concatenate([env.qpos, env.qvel, env.warm, env.sens, [img.ncon, img.nefc, img.niter]]))
    res[name] = np.array(rec)
    print(name, "ncon max", int(res[name][:, -3].max()), "niter max", int(res[name][:, -1].max()), flush=True)
np.savez(out, **res)
