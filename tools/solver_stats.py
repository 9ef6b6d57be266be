"""Distribution of contacts / rows / PGS sweeps over the batch in the bench regime (a new random action every step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
h = _capi.Handle(blob.pack(m), n_env)
h.reset()
rng = np.random.default_rng(0)
ioff = h.lds_offset("ints")
info_at = h.lds_offset("i_rowinfo")
for t in range(1001):
    h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
    if t % 100 == 0 and t > 0:
        img = h.step_debug(None, 0, 1, 0)
        ints = img[:, ioff:ioff + (info_at + m.njmax + 1) // 2 + 1].copy().view(np.int32)
        ncon, nefc, niter = ints[:, 0], ints[:, 1], ints[:, 3]
        tree = (ints[:, info_at:info_at + m.njmax] >> 19) - 2
        valid = np.arange(m.njmax)[None, :] < nefc[:, None]
        per_tree = np.stack([((tree == k) & valid).sum(1) for k in range(m.ntree)], 1)
        tmax = per_tree.max(1)
        steps = tmax * niter
        print(f"step {t}: ncon mean {ncon.mean():.2f} max {ncon.max()} | nefc mean {nefc.mean():.1f} max {nefc.max()} | rows/tree max mean {tmax.mean():.1f} "
              f"max {tmax.max()} | sweeps mean {niter.mean():.1f} p50 {np.percentile(niter,50):.0f} p90 {np.percentile(niter,90):.0f} p99 {np.percentile(niter,99):.0f} "
              f"max {niter.max()} | row steps (rows/tree x sweeps) mean {steps.mean():.0f} p99 {np.percentile(steps,99):.0f} max {steps.max()}", flush=True)
    else:
        h.step_device(None, 0, 1)
