"""When do the waves of one step launch run?  Per-wave start / end times (constant 100 MHz clock) of a launch in the
contact regime: how full the machine is over the launch, how long a wave takes, what the last waves to finish are."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
os.environ["MJRL_SPEC_FLAGS"] = (os.environ.get("MJRL_SPEC_FLAGS", "") + " -DMJRL_DIAG").strip()   # the diagnostic build of the specialised kernel
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi

n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
level = sys.argv[2] if len(sys.argv) > 2 else "two_agent.xml"
m = mjcf.compile_mjcf(levels.level_path(level))
h = _capi.Handle(blob.pack(m), n_env)
h.reset()
h.set_scatter_tables([list(range(m.nu))], 0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
ring = torch.from_numpy(rng.uniform(-1, 1, (64, n_env, m.nu))).to(dev)
ioff = h.lds_offset("ints")
slots = 256 * min(8, int(160 * 1024 // ((h.size("lds_doubles") * 8 + 1279) // 1280 * 1280)))      # wave slots of the chip for this image
for t in range(400):
    h.step_device(ring[t % 64].data_ptr(), m.nu, 1)
prev = None
h2 = _capi.Handle(blob.pack(m), n_env)          # replays the step with a debug dump: rows and sweeps of every copy
h2.reset()
h2.set_scatter_tables([list(range(m.nu))], 0)
info_at = h.lds_offset("i_rowinfo")
for rep in range(3):
    for f in ("qpos", "qvel", "qacc_warmstart", "ctrl"):
        h2.set_field(f, h.get_field(f))
    img = h2.step_debug(ring[(400 + rep) % 64].data_ptr(), m.nu, 1, 0)
    ints = img[:, ioff:ioff + (info_at + m.njmax + 1) // 2 + 1].copy().view(np.int32)
    nefc, niter = ints[:, 1], ints[:, 3]
    trees = (ints[:, info_at:info_at + m.njmax] >> 19) - 2
    valid = np.arange(m.njmax)[None, :] < nefc[:, None]
    per_tree = np.stack([((trees == k) & valid).sum(1) for k in range(m.ntree)], 1).max(1)
    tl = h.step_timeline(ring[(400 + rep) % 64].data_ptr(), m.nu).astype(np.int64)
    start, end, env = tl[:, 0], tl[:, 1], tl[:, 2]
    t0 = start.min()
    start, end = (start - t0) / 100.0, (end - t0) / 100.0          # microseconds
    dur = end - start
    total = end.max()
    print(f"launch {rep}: {total:.1f} us from first wave start to last wave end; wave duration mean {dur.mean():.1f} "
          f"p50 {np.percentile(dur, 50):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} us; "
          f"sum of durations / ({slots} slots x launch) = {dur.sum() / (slots * total):.2f}")
    edges = np.linspace(0, total, 11)
    occ = [(np.minimum(end, b) - np.maximum(start, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("  waves in flight per tenth of the launch: " + " ".join(f"{o:6.0f}" for o in occ))
    order = np.argsort(start)
    k = n_env // 8
    print("  by dispatch order (eighths): start " + " ".join(f"{start[order[i * k:(i + 1) * k]].mean():6.1f}" for i in range(8)))
    print("                            duration " + " ".join(f"{dur[order[i * k:(i + 1) * k]].mean():6.1f}" for i in range(8)))
    last = np.argsort(end)[-5:]
    two_tree = ((trees == -2) & valid).sum(1)
    print(f"  copies with rows that couple two trees (schedule solver): {(two_tree > 0).sum()}, their wave duration mean "
          f"{dur[np.isin(env, np.nonzero(two_tree > 0)[0])].mean() if (two_tree > 0).any() else 0:.1f} us; copies with 17+ rows in a tree: {(per_tree > 16).sum()}")
    print("  last five waves: " + "; ".join(f"wg {w} start {start[w]:.1f} dur {dur[w]:.1f} rows/tree {per_tree[env[w]]} sweeps {niter[env[w]]}" for w in last))
    wide = per_tree[env] > 16
    print(f"  waves with 17+ rows in a tree (wide solver): {wide.sum()}, duration mean {dur[wide].mean() if wide.any() else 0:.1f} max {dur[wide].max() if wide.any() else 0:.1f}; "
          f"of the 20 longest waves {wide[np.argsort(dur)[-20:]].sum()} are wide; 16-row solver waves with 60+ sweeps: {((~wide) & (niter[env] >= 60)).sum()}, "
          f"duration mean {dur[(~wide) & (niter[env] >= 60)].mean():.1f}")
    # how well does the previous step predict this one?  (the heaviest 2 % of the waves)
    d_env = np.zeros(n_env); d_env[env] = dur
    pos_env = np.zeros(n_env, np.int64); pos_env[env[order]] = np.arange(n_env)
    if prev is not None:
        heavy = np.argsort(d_env)[-n_env // 50:]
        rank_prev = np.argsort(np.argsort(prev))[heavy]
        print(f"  heaviest 2% now: previous-step duration rank median {np.median(rank_prev):.0f} min {rank_prev.min()} of {n_env}; "
              f"dispatch position median {np.median(pos_env[heavy]):.0f} max {pos_env[heavy].max()}; "
              f"in the second round: {(pos_env[heavy] >= 2048).sum()} of {heavy.size}; corr(prev dur, dur) {np.corrcoef(prev, d_env)[0, 1]:.2f}")
    prev = d_env
h.close()
