"""Declared plugin vocabulary that the step kernel can run on the device ("GPU fast path").

Every class here is an ordinary reference-style plugin -- an environmentDynamics class with ``dynamic(agent, actions)
-> (reward, obs, done, info)`` (mujoco_rl.py:124, 236) or a ``f(env, agent)`` reward / done callable
(mujoco_rl.py:158-169, 145-156) -- so it also runs in the host plugin loop.  In addition each one describes itself as
an op of the fused program (``fused_op``); when *all* configured plugins do, ``MuJoCoRL`` uploads the program with
``mjrl_set_program`` and a ``step()`` is one kernel launch with no host plugin loop.  The two paths are compared in
tests/test_gpu_parity.py.

* ``Language``            -- the README's language channel (README.md:109-136) in its 4-tuple form.
* ``TargetDistanceReward`` -- negative distance, or distance decrease since the last step, between the agent's body and a
  named body / geom (the target-seeking reward of the reference's tutorial notebook and Testing/EnvironmentDynamic.py).
* ``TargetReached``       -- done when that distance falls below a threshold.
"""
from __future__ import annotations

import numpy as np

OP_LANGUAGE, OP_DIST_REWARD, OP_DIST_DONE = 1, 2, 3


class Language:
    """Each agent utters ``int(action)``; its observation is what the other agent last uttered (0 before it spoke)."""

    def __init__(self, mujoco_gym):
        self.mujoco_gym = mujoco_gym
        self.observation_space = {"low": [0], "high": [3]}
        self.action_space = {"low": [0], "high": [3]}

    def dynamic(self, agent, actions):
        env = self.mujoco_gym
        store = env.data_store
        actions = np.asarray(actions)
        utterance = np.trunc(actions[..., 0]).astype(np.int64)
        store[agent]["utterance"] = int(utterance) if utterance.ndim == 0 else utterance
        other = [name for name in env.agents if name != agent][0]
        heard = store[other].get("utterance", 0)
        heard = np.asarray(heard, dtype=np.float64)
        obs = heard.reshape(1) if heard.ndim == 0 else heard.reshape(-1, 1)
        return 0, obs, (False if utterance.ndim == 0 else np.zeros(utterance.shape, bool)), {}

    def fused_op(self, env, alloc):
        lo, hi = env.action_routing["dynamic"][self.__class__.__name__]
        return dict(kind=OP_LANGUAGE, i=[lo, alloc.slot("utterance"), alloc.extra_obs(1)], f=[])


class _Target:
    def __init__(self, target: str):
        self.target = target

    def _target_ref(self, env):
        names = env._compiled.names
        if self.target in names["body"]:
            return 0, names["body"].index(self.target)
        if self.target in names["geom"]:
            return 1, names["geom"].index(self.target)
        raise Exception(f"target '{self.target}' is neither a body nor a geom of the level")


class TargetDistanceReward(_Target):
    """``mode="negative"``: reward = -scale * distance.  ``mode="delta"``: reward = scale * (previous distance - distance),
    0 on the first step of an episode; the previous distance lives in the agent's data store under ``key``."""

    def __init__(self, target: str, mode: str = "negative", scale: float = 1.0, key: str = "distance"):
        super().__init__(target)
        if mode not in ("negative", "delta"):
            raise Exception("mode must be 'negative' or 'delta'")
        self.mode, self.scale, self.key = mode, float(scale), key

    def __call__(self, env, agent):
        dist = env.distance(agent, self.target)
        store = env.data_store[agent]
        if self.mode == "negative":
            reward = self.scale * (-dist)
        else:
            prev = store.get(self.key)
            reward = 0.0 * dist if prev is None else self.scale * (prev - dist)
        store[self.key] = dist
        return float(reward) if np.ndim(reward) == 0 else reward

    def fused_op(self, env, alloc):
        kind, ident = self._target_ref(env)
        return dict(kind=OP_DIST_REWARD, i=[kind, ident, alloc.slot(self.key), 0 if self.mode == "negative" else 1],
                    f=[self.scale])


class TargetReached(_Target):
    def __init__(self, target: str, threshold: float):
        super().__init__(target)
        self.threshold = float(threshold)

    def __call__(self, env, agent):
        dist = env.distance(agent, self.target)
        done = dist < self.threshold
        return bool(done) if np.ndim(done) == 0 else done

    def fused_op(self, env, alloc):
        kind, ident = self._target_ref(env)
        return dict(kind=OP_DIST_DONE, i=[kind, ident], f=[self.threshold])


class ProgramBuilder:
    """Allocates data-store slots and extra-observation indices while the plugins describe their ops."""

    def __init__(self):
        self.slots, self.n_extra, self.ops = {}, 0, []

    def slot(self, key: str) -> int:
        return self.slots.setdefault(key, len(self.slots))

    def extra_obs(self, width: int) -> int:
        first = self.n_extra
        self.n_extra += width
        return first

    def add(self, op: dict):
        self.ops.append(op)

    def arrays(self):
        pi = np.zeros((len(self.ops), 8), np.int32)
        pf = np.zeros((len(self.ops), 4), np.float64)
        for k, op in enumerate(self.ops):
            pi[k, 0] = op["kind"]
            pi[k, 1:1 + len(op["i"])] = op["i"]
            pf[k, :len(op["f"])] = op["f"]
        return pi, pf


def build_program(env):
    """The fused program for ``env``'s plugins, or ``None`` when any of them is outside the vocabulary.
    Order: dynamics in list order, then reward functions, then done functions (mujoco_rl.py:268-286)."""
    plugins = list(env.environment_dynamics) + list(env.reward_functions) + list(env.done_functions)
    if not plugins or not all(hasattr(p, "fused_op") for p in plugins):
        return None
    builder = ProgramBuilder()
    for plugin in plugins:
        builder.add(plugin.fused_op(env, builder))
    return builder
