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
* ``TargetDynamic``       -- target switching among the objects that carry a tag of the info JSON
  (Testing/EnvironmentDynamic.py:17-32); ``PickUpDynamic`` -- the same with the inventory toggle and reward of
  Testing/Pick_Up_Dynamic.py:15-41.  The two reward / done classes above also take ``current_target_of=<tag>``: the
  agent's current target (Testing/SingleAgentTest.py:41-48).  Random choices come from ``mix64``, the counter-based
  generator the device uses, so the host loop and the fused ops draw the same targets.
"""
from __future__ import annotations

import numpy as np

OP_LANGUAGE, OP_DIST_REWARD, OP_DIST_DONE, OP_TARGET = 1, 2, 3, 4

_M = (1 << 64) - 1


def mix64(seed, env, agent, step, salt):
    """The counter-based generator of the device's random choices (csrc/mjrl_step.h ``mix64``), bit for bit: a pure function
    of (seed, global copy id, agent index, step key, salt).  ``env`` and ``step`` may be arrays.  The plugins' step key is
    ``episode_key(env)``: the copy's episode count in the upper and its episode step in the lower 32 bits, so that a
    copy's episodes draw different targets (the reference draws with random.randint)."""
    env = np.asarray(env, dtype=np.uint64)
    step = np.asarray(step, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(int(seed) * 0x9E3779B97F4A7C15 & _M) + env * np.uint64(0xBF58476D1CE4E5B9)
             + np.uint64(int(agent) * 0x94D049BB133111EB & _M) + step * np.uint64(0xD6E8FEB86659FD93)
             + np.uint64(int(salt) * 0xA0761D6478BD642F & _M))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def episode_key(env):
    """``episode << 32 | episode step`` of every copy of ``env`` (uint64 array): the step key of the device's draws."""
    episode = np.asarray(getattr(env, "episode", 0), dtype=np.uint64).reshape(-1)
    return (episode << np.uint64(32)) | np.uint64(int(env.timestep))


def pick_of(z, n: int):
    return ((np.asarray(z, np.uint64) >> np.uint64(33)) % np.uint64(max(n, 1))).astype(np.int64)


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


class TargetDynamic:
    """Target seeking (Testing/EnvironmentDynamic.py:17-32, in the 4-tuple form of mujoco_rl.py:236): the objects tagged
    ``tag`` in the level's info JSON are the targets; every agent has a current one (``data_store[agent]["current_target"]``,
    its index in ``filter_by_tag(tag)``), chosen at random when the episode starts and again whenever the agent comes
    within ``threshold`` of it; the observation is the current target's position.  Subclass to change ``tag``,
    ``threshold``, ``seed``.  The random choices come from ``mix64`` keyed on (seed, global copy id, agent, episode count and
    episode step), so the host loop and the fused op draw the same targets, and every episode of a copy its own."""
    tag, threshold, reach_reward, inventory, seed = "target", 1.0, 0.0, False, 0

    def __init__(self, mujoco_gym):
        self.mujoco_gym = mujoco_gym
        self.observation_space = {"low": [-70, -70, -70] + ([0] if self.inventory else []),
                                  "high": [70, 70, 70] + ([1] if self.inventory else [])}
        self.action_space = {"low": [], "high": []}

    def dynamic(self, agent, actions):
        env = self.mujoco_gym
        store = env.data_store[agent]
        batched = env.n_env > 1
        ids = env.first_env_id + np.arange(env.n_env)
        k = env.agents.index(agent)
        if "targets" not in store:
            store["targets"] = env.tagged_names(self.tag)
        names = store["targets"]
        if "current_target" not in store:
            store["current_target"] = pick_of(mix64(self.seed, ids, k, episode_key(env), 0), len(names))
            if self.inventory:
                store["inventory"] = np.zeros(env.n_env)
        cur = np.asarray(store["current_target"]).reshape(env.n_env)
        places = np.stack([np.asarray(env.get_data(name)["position"], np.float64).reshape(env.n_env, 3) for name in names], axis=1)
        here = np.asarray(env.data.body(agent).xipos, np.float64).reshape(env.n_env, 3)
        rows = np.arange(env.n_env)
        dist = np.linalg.norm(here - places[rows, cur], axis=1)
        reached = dist < self.threshold
        reward = np.where(reached, self.reach_reward, 0.0)
        if self.inventory:
            inv = np.asarray(store["inventory"], np.float64).reshape(env.n_env)
            store["inventory"] = np.where(reached, 1.0 - inv, inv)
        fresh = pick_of(mix64(self.seed, ids, k, episode_key(env), 1), len(names))
        cur = np.where(reached, fresh, cur)
        store["current_target"] = cur
        if reached.any():
            old = np.asarray(store.get("distance", np.full(env.n_env, np.nan)), np.float64).reshape(env.n_env)
            store["distance"] = np.where(reached, np.linalg.norm(here - places[rows, cur], axis=1), old)
        obs = places[rows, cur]
        if self.inventory:
            obs = np.concatenate([obs, np.asarray(store["inventory"]).reshape(env.n_env, 1)], axis=1)
        if not batched:
            return float(reward[0]), obs[0], False, {}
        return reward, obs, np.zeros(env.n_env, bool), {}

    def fused_op(self, env, alloc):
        return dict(kind=OP_TARGET, i=[alloc.tag(self.tag), alloc.slot("current_target"),
                                       alloc.slot("inventory") if self.inventory else -1,
                                       alloc.extra_obs(4 if self.inventory else 3), alloc.slot("distance")],
                    f=[self.threshold, self.reach_reward, float(self.seed)])


class PickUpDynamic(TargetDynamic):
    """The pick-up toggle of Testing/Pick_Up_Dynamic.py:15-41: reaching the current target (distance < 2) flips the
    agent's inventory between 0 and 1, pays reward 1 and selects a new target; the observation is the target's position
    followed by the inventory."""
    threshold, reach_reward, inventory = 2.0, 1.0, True


class _Target:
    """``target``: name of a body or geom -- or ``current_target_of=tag``: the agent's current target among the objects
    tagged ``tag`` (kept by a ``TargetDynamic`` / ``PickUpDynamic``; Testing/SingleAgentTest.py:41-48)."""

    def __init__(self, target: str | None, current_target_of: str | None = None):
        if (target is None) == (current_target_of is None):
            raise Exception("give either a target name or current_target_of=<tag>")
        self.target, self.tag = target, current_target_of

    def _target_ref(self, env, alloc):
        """(kind, id, store slot of the current target's index or 0)."""
        if self.tag is not None:
            return 2, alloc.tag(self.tag), alloc.slot("current_target")
        names = env._compiled.names
        if self.target in names["body"]:
            return 0, names["body"].index(self.target), 0
        if self.target in names["geom"]:
            return 1, names["geom"].index(self.target), 0
        raise Exception(f"target '{self.target}' is neither a body nor a geom of the level")

    def _distance(self, env, agent):
        """Distance to the target; None while there is no current target."""
        if self.tag is None:
            return env.distance(agent, self.target)
        store = env.data_store[agent]
        if "current_target" not in store:
            return None
        names = env.tagged_names(self.tag)
        cur = np.asarray(store["current_target"]).reshape(env.n_env)
        places = np.stack([np.asarray(env.get_data(name)["position"], np.float64).reshape(env.n_env, 3) for name in names], axis=1)
        here = np.asarray(env.data.body(agent).xipos, np.float64).reshape(env.n_env, 3)
        dist = np.linalg.norm(here - places[np.arange(env.n_env), cur], axis=1)
        return float(dist[0]) if env.n_env == 1 else dist


class TargetDistanceReward(_Target):
    """``mode="negative"``: reward = -scale * distance.  ``mode="delta"``: reward = scale * (previous distance - distance),
    0 on the first step of an episode; the previous distance lives in the agent's data store under ``key``."""

    def __init__(self, target: str | None = None, mode: str = "negative", scale: float = 1.0, key: str = "distance",
                 current_target_of: str | None = None):
        super().__init__(target, current_target_of)
        if mode not in ("negative", "delta"):
            raise Exception("mode must be 'negative' or 'delta'")
        self.mode, self.scale, self.key = mode, float(scale), key

    def __call__(self, env, agent):
        dist = self._distance(env, agent)
        if dist is None:
            return 0.0 if env.n_env == 1 else np.zeros(env.n_env)
        store = env.data_store[agent]
        if self.mode == "negative":
            reward = self.scale * (-dist)
        else:
            prev = store.get(self.key)
            if prev is None:
                reward = 0.0 * dist
            else:       # (with several copies the key may be set for some of them only: NaN = absent, as on the device)
                prev = np.asarray(prev, dtype=np.float64)
                reward = np.where(np.isnan(prev), 0.0, self.scale * (prev - dist))
        store[self.key] = dist
        return float(reward) if np.ndim(reward) == 0 else reward

    def fused_op(self, env, alloc):
        kind, ident, cur = self._target_ref(env, alloc)
        return dict(kind=OP_DIST_REWARD, i=[kind, ident, alloc.slot(self.key), 0 if self.mode == "negative" else 1, cur],
                    f=[self.scale])


class TargetReached(_Target):
    def __init__(self, target: str | None = None, threshold: float = 1.0, current_target_of: str | None = None):
        super().__init__(target, current_target_of)
        self.threshold = float(threshold)

    def __call__(self, env, agent):
        dist = self._distance(env, agent)
        if dist is None:
            return False if env.n_env == 1 else np.zeros(env.n_env, bool)
        done = dist < self.threshold
        return bool(done) if np.ndim(done) == 0 else done

    def fused_op(self, env, alloc):
        kind, ident, cur = self._target_ref(env, alloc)
        return dict(kind=OP_DIST_DONE, i=[kind, ident, 0, 0, cur], f=[self.threshold])


class ProgramBuilder:
    """Allocates data-store slots and extra-observation indices while the plugins describe their ops."""

    def __init__(self):
        self.slots, self.n_extra, self.ops, self.tags = {}, 0, [], []

    def tag(self, name: str) -> int:
        """Index of an info-JSON tag in the device tag tables (mjrl_set_tag_tables)."""
        if name not in self.tags:
            self.tags.append(name)
        return self.tags.index(name)

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
