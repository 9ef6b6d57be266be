"""Which history of a copy's solver work best predicts that it will be among the heaviest of the next launch?
(The launch ends with the heavy waves that were dispatched late.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
work, rows, cons = [], [], []
for t in range(480):
    h.set_field("ctrl", rng.uniform(-1, 1, (n_env, m.nu)))
    if t >= 400:
        img = h.step_debug(None, 0, 1, 0)
        ints = img[:, ioff:ioff + 4].copy().view(np.int32)
        work.append(ints[:, 1].astype(float) * ints[:, 3]); rows.append(ints[:, 1].astype(float)); cons.append(ints[:, 0].astype(float))
    else:
        h.step_device(None, 0, 1)
W, R, C = np.array(work), np.array(rows), np.array(cons)
N = W / np.maximum(R, 1)            # sweeps
def bucket(w):                       # the kernel files copies under log2 buckets
    return np.floor(np.log2(np.maximum(w, 0.5))) + 1
preds = {
    "work(t-1)": lambda t: W[t - 1],
    "log2 bucket of work(t-1) [the kernel today]": lambda t: bucket(W[t - 1]),
    "max of last 2": lambda t: np.maximum(W[t - 1], W[t - 2]),
    "max of last 4": lambda t: W[t - 4:t].max(0),
    "max of last 8": lambda t: W[t - 8:t].max(0),
    "decayed max (0.7/step) of last 8": lambda t: max_decay(t, 0.7),
    "decayed max (0.85/step) of last 8": lambda t: max_decay(t, 0.85),
    "bucket of decayed max 0.85": lambda t: bucket(max_decay(t, 0.85)),
    "mean of last 4": lambda t: W[t - 4:t].mean(0),
    "rows(t-1)": lambda t: R[t - 1],
    "contacts(t-1), then work": lambda t: C[t - 1] * 1e6 + W[t - 1],
    "rows x (sweeps + 10)": lambda t: R[t - 1] * (N[t - 1] + 10),
    "rows x (sweeps + 30)": lambda t: R[t - 1] * (N[t - 1] + 30),
    "rows^2 x (sweeps + 10)": lambda t: R[t - 1] ** 2 * (N[t - 1] + 10),
    "rows x max sweeps of last 4": lambda t: R[t - 1] * N[t - 4:t].max(0),
    "max(work(t-1), 8 x rows(t-1))": lambda t: np.maximum(W[t - 1], 8 * R[t - 1]),
    "max(work(t-1), 20 x rows(t-1))": lambda t: np.maximum(W[t - 1], 20 * R[t - 1]),
}
def max_decay(t, d):
    return np.max([W[t - k] * d ** (k - 1) for k in range(1, 9)], 0)
for name, f in preds.items():
    late, worst = [], []
    for t in range(8, len(W)):
        p = f(t) + 1e-9 * np.arange(n_env)            # ties: index order
        rank = np.argsort(np.argsort(-p))
        top = np.argsort(-W[t])[:64]
        late.append((rank[top] >= 1536).sum()); worst.append(rank[top].max())
        late2 = late2 + [(rank[top] >= 2048).sum()] if t > 8 else [(rank[top] >= 2048).sum()]
    print(f"{name:48s}: of the 64 heaviest, dispatched after position 1536: mean {np.mean(late):5.2f}  after 2048: {np.mean(late2):5.2f}  worst position: mean {np.mean(worst):6.0f}")
