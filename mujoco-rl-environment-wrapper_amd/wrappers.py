"""Adapters above the step path.

* ``GymnasiumWrapper`` -- the reference's single-agent shim (MuJoCo_Gym/wrappers.py:12-82): same constructor, same
  ``step`` / ``reset`` return values; it subclasses ``gymnasium.Env`` when gymnasium is installed.
* ``BatchedVectorEnv`` -- what RL libraries consume once the batch lives on one device: a single-agent vector env in
  the Gymnasium ``VectorEnv`` calling convention (``num_envs``, batched arrays, automatic reset of finished
  copies), over ``MuJoCoRL(numEnvs=N)``.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium
    _EnvBase = gymnasium.Env
except Exception:
    _EnvBase = object


class GymnasiumWrapper(_EnvBase):
    metadata = {"render_modes": ["human", "none"], "render_fps": 4}

    def __init__(self, environment, agent: str, render_mode="none") -> None:
        if _EnvBase is not object:
            super().__init__()
        self.environment = environment
        self.agent = agent
        self.render_mode = render_mode
        if len(self.environment.agents) > 1:
            raise Exception("Environment has too many agents. Only one agent is allowed in a gym environment.")
        self.observation_space = environment.observation_space(agent)
        self.action_space = environment.action_space(agent)

    def step(self, action):
        observations, rewards, terminations, truncations, infos = self.environment.step({self.agent: action})
        return (observations[self.agent], rewards[self.agent], terminations[self.agent], truncations["__all__"],
                infos[self.agent])

    def reset(self, *, seed=1, options={}):
        observations, infos = self.environment.reset()
        return observations[self.agent], infos

    def render(self):
        pass


class BatchedVectorEnv:
    """``num_envs`` copies of a single-agent level behind the vector-env calling convention.

    ``step(actions[num_envs, act_dim]) -> (obs, rewards, terminations, truncations, infos)``; a copy whose episode
    ended (terminated or truncated) is reset before the next step and its fresh observation is returned in place of
    the terminal one, with the terminal observation under ``infos["final_observation"]`` (Gymnasium autoreset)."""

    def __init__(self, environment, agent: str | None = None):
        if len(environment.agents) != 1 and agent is None:
            raise Exception("BatchedVectorEnv drives one agent; pass `agent` for a multi-agent level")
        self.environment = environment
        self.agent = agent or environment.agents[0]
        self.num_envs = environment.n_env
        self.single_observation_space = environment.observation_space(self.agent)
        self.single_action_space = environment.action_space(self.agent)
        self._steps = np.zeros(self.num_envs, np.int64)

    def _others(self, n):
        env = self.environment
        return {a: np.zeros((n,) + env.action_space(a).shape) for a in env.agents if a != self.agent}

    def reset(self, *, seed=None, options=None):
        observations, infos = self.environment.reset()
        self._steps[:] = 0
        return np.atleast_2d(observations[self.agent]), infos

    def step(self, actions):
        env = self.environment
        act = {self.agent: np.asarray(actions, dtype=np.float64).reshape(self.num_envs, -1), **self._others(self.num_envs)}
        observations, rewards, terminations, _, infos = env.step(act)
        obs = np.atleast_2d(observations[self.agent]).copy()
        self._steps += 1
        terminated = np.atleast_1d(terminations[self.agent]).astype(bool)
        truncated = self._steps > env.max_steps            # per-copy horizon (mujoco_rl.py:412: call max_steps + 1)
        done = terminated | truncated
        info = {"agent": infos[self.agent]}
        if done.any():
            info["final_observation"] = obs.copy()
            env._handle.reset(done.astype(np.uint8))
            env._obs_cache = None
            fresh = np.atleast_2d(env.get_observations(self.agent))
            width = min(fresh.shape[1], obs.shape[1])
            obs[done, :width] = fresh[done, :width]
            self._steps[done] = 0
        return obs, np.atleast_1d(rewards[self.agent]).astype(np.float64), terminated, truncated, info

    def close(self):
        self.environment.close()
