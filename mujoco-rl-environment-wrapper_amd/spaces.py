"""``Box`` space for the action / observation spaces (mujoco_rl.py:191-192, 211-212 build gymnasium Boxes).

gymnasium is used when it is installed; otherwise this stand-in offers the part of its interface the
reference relies on (``low``, ``high``, ``shape``, ``dtype``, ``sample()``, ``contains()``).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box  # noqa: F401
except Exception:
    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            self.low = np.asarray(low, dtype=self.dtype)
            self.high = np.asarray(high, dtype=self.dtype)
            if self.low.shape != self.high.shape:
                raise ValueError("low and high must have the same shape")
            self.shape = self.low.shape if shape is None else tuple(shape)
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            low = np.where(np.isfinite(self.low), self.low, -1e6)
            high = np.where(np.isfinite(self.high), self.high, 1e6)
            return self._rng.uniform(low, high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"
