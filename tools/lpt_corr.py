import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
n_env = 4096
m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
h = _capi.Handle(blob.pack(m), n_env)
h.reset()
rng = np.random.default_rng(0)
ioff = h.lds_offset("ints")
prev = None
hist = []
for t in range(440):
    h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
    if t >= 400:
        img = h.step_debug(None, 0, 1, 0)
        ints = img[:, ioff:ioff + 4].copy().view(np.int32)
        nefc, niter = ints[:, 1].astype(float), ints[:, 3].astype(float)
        hist.append((nefc, niter))
    else:
        h.step_device(None, 0, 1)
for k in range(1, 6):
    w0 = hist[k - 1][0] * hist[k - 1][1]; w1 = hist[k][0] * hist[k][1]
    top = np.argsort(-w1)[:64]
    rank_pred = np.argsort(np.argsort(-w0))
    print(f"step {k}: corr(work) {np.corrcoef(w0, w1)[0,1]:.2f}  corr(nefc) {np.corrcoef(hist[k-1][0], hist[k][0])[0,1]:.2f}  corr(niter) {np.corrcoef(hist[k-1][1], hist[k][1])[0,1]:.2f} "
          f"| of the 64 heaviest now: median predicted rank {np.median(rank_pred[top]):.0f}, worst {rank_pred[top].max()}, beyond 1536: {(rank_pred[top] >= 1536).sum()}")
