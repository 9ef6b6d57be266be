"""Where do the generic and the model-specialised step kernel stop agreeing bit for bit?  One step from the reset state
(and after a few settling steps) with a dump of the copy's LDS image from both builds; regions compared in stage order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
name = sys.argv[1] if len(sys.argv) > 1 else "two_agent.xml"
m = mjcf.compile_mjcf(levels.level_path(name))
packed = blob.pack(m)
n = 8
spec, gen = _capi.Handle(packed, n, specialize=True), _capi.Handle(packed, n, specialize=False)
print(spec.kernel, gen.kernel)
regions = ["qpos", "qvel", "xpos", "xquat", "com", "cdof", "LD", "Dinv", "smooth", "qaccs", "qfc", "qacc", "con", "sens", "J", "row"]
offs = {r: spec.lds_offset(r) for r in regions}
order = sorted(regions, key=lambda r: offs[r])
total = spec.size("lds_doubles")
rng = np.random.default_rng(1)
for h in (spec, gen):
    h.reset()
for step in range(int(sys.argv[2]) if len(sys.argv) > 2 else 260):
    ctrl = rng.uniform(-1, 1, (n, m.nu))
    imgs = []
    for h in (spec, gen):
        h.set_field("ctrl", ctrl)
        imgs.append(h.step_debug(None, 0, 1, 0))
    a, b = imgs
    if not np.array_equal(a, b):
        print("step", step, "images differ")
        for i, r in enumerate(order):
            lo = offs[r]
            hi = min([offs[x] for x in order if offs[x] > lo] + [total])
            d = np.abs(a[:, lo:hi] - b[:, lo:hi])
            bad = np.argwhere(~((a[:, lo:hi] == b[:, lo:hi]) | (np.isnan(a[:, lo:hi]) & np.isnan(b[:, lo:hi]))))
            if len(bad):
                e, k = bad[0]
                print(f"  {r:8s} [{lo}:{hi}] {len(bad)} differing, first copy {e} index {k}: {a[e, lo + k]!r} vs {b[e, lo + k]!r}")
        break
else:
    print("no difference")
