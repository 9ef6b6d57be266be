"""Double-buffered halves: the batch as two handles of n/2 copies on two HIP streams, each stepping on its own
(the tail of one half's launch overlaps the body of the other's), against one handle of n copies in lockstep.
Random controls are written every step in both cases, from the same stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi

n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
packed = blob.pack(m)
dev = torch.device("cuda:0")


def run(parts):
    per = n_env // parts
    hs = [_capi.Handle(packed, per) for _ in range(parts)]
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    rng = np.random.default_rng(0)
    ring = [torch.from_numpy(rng.uniform(-1, 1, (64, per, m.nu))).to(dev) for _ in range(parts)]
    scatter = [list(range(m.nu))]          # one "agent" owning every actuator
    for h, s in zip(hs, streams):
        h.set_stream(s.cuda_stream)
        h.reset()
        h.set_scatter_tables(scatter, 0)
    torch.cuda.synchronize(dev)
    t0 = None
    for t in range(300 + steps):
        if t == 300:
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for h, r in zip(hs, ring):
            h.step_device(r[t % 64].data_ptr(), m.nu, 1)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    for h in hs:
        h.close()
    return dt


for parts in (1, 2, 4):
    dt = run(parts)
    print(f"{parts} handle(s) x {n_env // parts} copies: {dt * 1e6:7.1f} us per step of all copies, {n_env / dt / 1e6:6.2f} M env-steps/s", flush=True)
