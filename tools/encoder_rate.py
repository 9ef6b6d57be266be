"""Config 5 with the encoder behind the cameras: step + both 64x64 cameras + autoencoder latents into the observation, per
step, device resident.  Usage: encoder_rate.py [n_env] [latent]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import levels
from mjrl_amd.mujoco_rl import MuJoCoRL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
latent = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(0)
w = {"w1": rng.standard_normal((3, 3, 3, 32)).astype(np.float32) * 0.2, "b1": np.zeros(32, np.float32),
     "w2": rng.standard_normal((3, 3, 32, 64)).astype(np.float32) * 0.08, "b2": np.zeros(64, np.float32),
     "wd": rng.standard_normal((16384, latent)).astype(np.float32) * 0.01, "bd": np.zeros(latent, np.float32)}
for enc in (False, True):
    cfg = {"xmlPath": levels.level_path("two_agent.xml"), "agents": ["sender", "receiver"], "numEnvs": n, "agentCameras": True}
    if enc:
        cfg["cameraEncoder"] = {"weights": w}
    env = MuJoCoRL(cfg)
    env.reset_batched()
    dev = torch.device("cuda", 0)
    acts = torch.from_numpy(rng.uniform(-1, 1, (64, n, 2, 8))).to(dev)
    rgb = torch.empty((n, 2, 64, 64, 3), dtype=torch.uint8, device=dev)
    bufs = None
    def step(i):
        global bufs
        bufs = env.step_batched(acts[i % 64], *(bufs or ()))
        if not enc:
            env._handle.render(64, 64, d_rgb=rgb.data_ptr())
    for i in range(300):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300, 500):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{'step + render + encode (latents in obs)' if enc else 'step + render':42s} {n} copies: {dt * 1e3:.3f} ms per step = {n / dt / 1e6:.3f} M env-steps/s")
    env.close()
