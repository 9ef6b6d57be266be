"""GPU bring-up: HIP path vs CPU oracle on a few env copies, then a raw throughput probe."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
from oracle.oracle import OracleEnv

name = sys.argv[1] if len(sys.argv) > 1 else "two_agent.xml"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nbig = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
m = mjcf.compile_mjcf(levels.level_path(name))
b = blob.pack(m)
n_env = 8
h = _capi.Handle(b, n_env)
print("created; lds doubles", h.size("lds_doubles"), "bytes", 8 * h.size("lds_doubles"), flush=True)
h.reset()
h.sync()
oras = [OracleEnv(b) for _ in range(n_env)]
rng = np.random.default_rng(0)
warm = h.get_field("qacc_warmstart")
print("warm diff after reset", np.abs(warm - np.stack([o.qacc_warmstart for o in oras])).max(), flush=True)
worst = 0
for step in range(nsteps):
    ctrl = rng.uniform(-1, 1, size=(n_env, m.nu))
    h.set_field("ctrl", ctrl)
    h.step_host(None, 1)
    qpos, qvel = h.get_field("qpos"), h.get_field("qvel")
    for e, o in enumerate(oras):
        o.ctrl[:] = ctrl[e]
        o.step()
    oq = np.stack([o.qpos for o in oras]); ov = np.stack([o.qvel for o in oras])
    rq = np.abs(qpos - oq).max() / max(1, np.abs(oq).max()); rv = np.abs(qvel - ov).max() / max(1, np.abs(ov).max())
    worst = max(worst, rq, rv)
    if step % 25 == 0 or not np.isfinite(rq):
        print(step, f"qpos rel {rq:.2e} qvel rel {rv:.2e}", "ncon(ora)", [o.ncon for o in oras], flush=True)
    if not np.isfinite(rq):
        break
print("worst rel", worst, flush=True)
h.close()

# throughput probe
h = _capi.Handle(b, nbig)
h.reset(); h.sync()
ctrl = rng.uniform(-1, 1, size=(nbig, m.nu)); h.set_field("ctrl", ctrl)
for _ in range(20):
    h.step_device(None, 0, 1)
h.sync()
t = time.perf_counter()
K = 100
for _ in range(K):
    h.step_device(None, 0, 1)
h.sync()
dt = time.perf_counter() - t
print(f"n_env {nbig}: {dt / K * 1e3:.3f} ms/step, {nbig * K / dt:.3e} env-steps/s", flush=True)
q = h.get_field("qpos")
print("finite", np.isfinite(q).all(), "z mean", q[:, 2].mean())
