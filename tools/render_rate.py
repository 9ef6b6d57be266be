"""Config 5: 512 copies of the 2-agent level, one physics step plus the two agent cameras (64x64 RGB each) per step,
everything device-resident.  Reports steps/s with and without the render and the render kernel's share."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = H = 64
m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
h = _capi.Handle(blob.pack(m), n_env)
h.reset()
h.set_scatter_tables([list(range(m.nu))], 0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ring = torch.from_numpy(rng.uniform(-1, 1, (64, n_env, m.nu))).to(dev)
ncam = h.size("ncam")
rgb = torch.empty((n_env, ncam, H, W, 3), dtype=torch.uint8, device=dev)
for with_render in (False, True):
    for t in range(300):
        h.step_device(ring[t % 64].data_ptr(), m.nu, 1)
    h.sync(); t0 = time.perf_counter()
    for t in range(300, 500):
        h.step_device(ring[t % 64].data_ptr(), m.nu, 1)
        if with_render:
            h.render(W, H, rgb.data_ptr())
    h.sync()
    dt = (time.perf_counter() - t0) / 200
    print(f"{n_env} copies, {ncam} cameras {W}x{H}, render {'on ' if with_render else 'off'}: {dt * 1e6:7.1f} us per step, "
          f"{n_env / dt / 1e6:5.2f} M env-steps/s" + (f", {n_env * ncam * W * H * 3 / dt / 1e9:.1f} GB/s of pixels" if with_render else ""), flush=True)
    h.reset()
print("non-black pixels in the last frame:", float((rgb.float().sum(-1) > 0).float().mean()))
