"""Encoder kernels alone: conv + dense on 1024 random images in HBM, HIP-event time per call (diagnostic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
latent = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(0)
w = {"w1": rng.standard_normal((3, 3, 3, 32)).astype(np.float32) * 0.2, "b1": np.zeros(32, np.float32),
     "w2": rng.standard_normal((3, 3, 32, 64)).astype(np.float32) * 0.08, "b2": np.zeros(64, np.float32),
     "wd": rng.standard_normal((16384, latent)).astype(np.float32) * 0.01, "bd": np.zeros(latent, np.float32)}
h = _capi.Handle(blob.pack(mjcf.compile_mjcf(levels.level_path("two_agent.xml"))), 2)
h.encoder_load(w)
dev = torch.device("cuda", 0)
img = torch.randint(0, 256, (n_img, 64, 64, 3), dtype=torch.uint8, device=dev)
lat = torch.empty((n_img, latent), dtype=torch.float32, device=dev)
h.set_stream(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(20):
    h.encode(d_rgb=img.data_ptr(), n_img=n_img, d_latent=lat.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    h.encode(d_rgb=img.data_ptr(), n_img=n_img, d_latent=lat.data_ptr())
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 200 * 1e3
flop = n_img * (2 * 1024 * 27 * 32 + 2 * 256 * 288 * 64 + 2 * 16384 * latent)
print(f"encode {n_img} images, latent {latent}: {us:.1f} us per call = {flop / us / 1e6:.1f} TFLOP/s ({flop / us / 1e6 / 2500 * 100:.1f} % of 2.5 PFLOP/s)")
