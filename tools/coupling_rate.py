"""How often does a copy of the 2-agent level hold constraint rows that couple the two trees (ant against ant: the
serial solver path) or 17+ rows in one tree (wide register solver)?  Sampled over an episode with random actions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
level = sys.argv[1] if len(sys.argv) > 1 else "two_agent.xml"
n_env = 4096
m = mjcf.compile_mjcf(levels.level_path(level))
h = _capi.Handle(blob.pack(m), n_env)
h.reset()
h.set_scatter_tables([list(range(m.nu))], 0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ring = torch.from_numpy(rng.uniform(-1, 1, (64, n_env, m.nu))).to(dev)
ioff, info_at = h.lds_offset("ints"), h.lds_offset("i_rowinfo")
samples = coupled = wide = launches_with_coupled = 0
for t in range(1000):
    if t >= 250 and t % 10 == 0:
        img = h.step_debug(ring[t % 64].data_ptr(), m.nu, 1, 0)
        ints = img[:, ioff:ioff + (info_at + m.njmax + 1) // 2 + 1].copy().view(np.int32)
        nefc = ints[:, 1]
        trees = (ints[:, info_at:info_at + m.njmax] >> 19) - 2
        valid = np.arange(m.njmax)[None, :] < nefc[:, None]
        c = (((trees == -2) & valid).sum(1) > 0).sum()
        w = (np.stack([((trees == k) & valid).sum(1) for k in range(m.ntree)], 1).max(1) > 16).sum()
        samples += 1; coupled += c; wide += w; launches_with_coupled += c > 0
    else:
        h.step_device(ring[t % 64].data_ptr(), m.nu, 1)
print(f"{level}: {samples} sampled launches of {n_env} copies (steps 250..1000): copies with coupling rows per launch {coupled / samples:.2f} "
      f"(launches with at least one: {launches_with_coupled}), copies with 17+ rows in a tree per launch {wide / samples:.2f}")
