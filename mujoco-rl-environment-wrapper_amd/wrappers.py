"""Adapters above the step path.

* ``GymnasiumWrapper`` -- the reference's single-agent shim (MuJoCo_Gym/wrappers.py:12-82): same constructor, same
  ``step`` / ``reset`` return values; it subclasses ``gymnasium.Env`` when gymnasium is installed.
* ``BatchedVectorEnv`` -- what RL libraries consume once the batch lives on one device: a single-agent vector env in
  the Gymnasium ``VectorEnv`` calling convention (``num_envs``, batched arrays, automatic reset of finished
  copies), over ``MuJoCoRL(numEnvs=N)``'s array path: numpy in / numpy out on the handle's pinned host buffers
  (``mjrl_step_pinned``), torch CUDA tensors in / out without leaving HBM (``mjrl_step_device``); the autoreset is kept
  by the step kernel itself (``mjrl_set_autoreset``), so a ``step()`` is one launch and nothing else.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium
    _EnvBase = gymnasium.Env
    _VecBase = gymnasium.vector.VectorEnv
except Exception:
    _EnvBase = object
    _VecBase = object


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


class BatchedVectorEnv(_VecBase):
    """``num_envs`` copies of a level behind the Gymnasium ``VectorEnv`` calling convention, one driven agent.

    ``reset() -> (obs[num_envs, obs_dim], infos)``; ``step(actions[num_envs, act_dim]) -> (obs, rewards, terminations,
    truncations, infos)``.  ``actions`` is a numpy array (outputs: numpy views of the handle's pinned host buffers, which
    the next step overwrites -- copy what has to survive it) or a torch CUDA tensor (outputs: torch tensors in HBM, the
    launch asynchronous on the current torch stream).  It is a ``gymnasium.vector.VectorEnv`` where gymnasium exists.

    Autoreset (``autoreset=``), kept on the device by ``mjrl_set_autoreset`` -- no mask is computed or moved by the host:

    * ``"next_step"`` (default; Gymnasium >= 1.0's ``AutoresetMode.NEXT_STEP``): the step in which a copy's episode ends
      returns the terminal observation and the flags; the NEXT ``step()`` resets that copy instead of stepping it (its
      action is ignored) and returns the first observation of the new episode, reward 0, flags clear.
    * ``"reset_then_step"``: the next ``step()`` resets the copy and applies the action in the same launch --
      ``env.reset(); env.step(a)`` of the reference's sampling loops (fps_benchmark.py:33-38); the reset observation
      itself is never returned.
    * ``"same_step"`` (SB3 ``VecEnv`` / Gymnasium < 1.0): the step in which the episode ends returns the first
      observation of the new episode, the terminal one under ``infos["final_observation"]`` (rows of the copies that
      ended; ``infos["_final_observation"]`` is the mask).  Costs a host look at the flags and a masked reset per step
      with an ended copy.

    Levels with several agents: ``agent`` names the driven one, the others receive zero actions (the reference's own
    ``GymnasiumWrapper`` refuses such levels, wrappers.py:21-22)."""

    MODES = {"next_step": 1, "reset_then_step": 2, "same_step": 0}

    def __init__(self, environment, agent: str | None = None, autoreset: str = "next_step"):
        if len(environment.agents) != 1 and agent is None:
            raise Exception("BatchedVectorEnv drives one agent; pass `agent` for a multi-agent level")
        if autoreset not in self.MODES:
            raise Exception(f"autoreset must be one of {sorted(self.MODES)}")
        if environment._program is None and (environment.environment_dynamics or environment.reward_functions or
                                             environment.done_functions):
            raise Exception("BatchedVectorEnv runs on the array path: host plugins are not called; use the fused "
                            "vocabulary (dynamics.py) or MuJoCoRL.step()")
        self.environment = environment
        self.agent = agent or environment.agents[0]
        self._k = environment.agents.index(self.agent)
        self.num_envs = environment.n_env
        self.autoreset = autoreset
        self.single_observation_space = environment.observation_space(self.agent)
        self.single_action_space = environment.action_space(self.agent)
        self.observation_space = _batch_space(self.single_observation_space, self.num_envs)
        self.action_space = _batch_space(self.single_action_space, self.num_envs)
        self._n_agent = len(environment.agents)
        self._act_dim = max(environment.action_space(a).shape[0] for a in environment.agents)
        self._width = self.single_observation_space.shape[0]
        self._torch_act = None
        environment._handle.set_autoreset(self.MODES[autoreset])

    # -- helpers
    def _numpy_actions(self, actions):
        """The driven agent's rows inside the pinned action buffer [num_envs, n_agent, act_dim]; the others stay 0."""
        env = self.environment
        if env._pinned is None or env._pinned[0] is not env._handle or env._pinned[1].shape[-1] != self._act_dim:
            env._pinned = (env._handle,) + env._handle.host_buffers(self._act_dim)
            env._pinned[1][:] = 0.0
        p_act = env._pinned[1]
        act = np.asarray(actions, dtype=np.float64).reshape(self.num_envs, -1)
        p_act[:, self._k, :act.shape[1]] = act
        return p_act

    def _torch_actions(self, actions):
        import torch
        if self._n_agent == 1 and actions.dim() == 2 and actions.shape[1] == self._act_dim and actions.dtype == torch.float64:
            return actions.reshape(self.num_envs, 1, self._act_dim)
        if self._torch_act is None or self._torch_act.device != actions.device:
            self._torch_act = torch.zeros((self.num_envs, self._n_agent, self._act_dim), dtype=torch.float64, device=actions.device)
        self._torch_act[:, self._k, :actions.shape[-1]] = actions.reshape(self.num_envs, -1)
        return self._torch_act

    # -- VectorEnv
    def reset(self, *, seed=None, options=None):
        env = self.environment
        env.reset_batched()
        obs = np.atleast_2d(env.get_observations(self.agent))
        full = np.zeros((self.num_envs, self._width))
        full[:, :obs.shape[1]] = obs              # (slots of fused dynamics read 0 after a reset, like the array path's)
        return full, {}

    def step(self, actions):
        env = self.environment
        k = self._k
        if isinstance(actions, np.ndarray) or not hasattr(actions, "data_ptr"):
            p_act = self._numpy_actions(actions)
            env._handle.step_pinned(self._act_dim, env.skip_frames)
            env.timestep += 1
            env._obs_cache = None
            _, _, obs, reward, term, trunc = env._pinned
            obs, reward, term, trunc = obs[:, k, :self._width], reward[:, k], term[:, k].view(np.bool_), trunc[:, k].view(np.bool_)
            info = {}
            if self.autoreset == "same_step":
                done = term | trunc
                if done.any():
                    info = {"final_observation": obs[done].copy(), "_final_observation": done.copy()}
                    env._handle.reset(done.astype(np.uint8))
                    fresh = np.atleast_2d(env.get_observations(self.agent))
                    obs = obs.copy()
                    obs[done, :fresh.shape[1]] = fresh[done]
                    obs[done, fresh.shape[1]:] = 0.0
            return obs, reward, term, trunc, info
        import torch
        full = env.step_batched(self._torch_actions(actions))
        obs, reward, term, trunc = full[0][:, k, :self._width], full[1][:, k], full[2][:, k].bool(), full[3][:, k].bool()
        info = {}
        if self.autoreset == "same_step":
            done = term | trunc
            if bool(done.any()):              # (a host look at the flags: this mode's price)
                info = {"final_observation": obs[done].clone(), "_final_observation": done.clone()}
                fresh = torch.empty_like(full[0])
                env._handle.reset_device(done.to(torch.uint8).contiguous().data_ptr(), fresh.data_ptr())
                obs = torch.where(done[:, None], fresh[:, k, :self._width], obs)
        return obs, reward, term, trunc, info

    def close(self, **kwargs):
        self.environment.close()


def _batch_space(space, n):
    """The batched Box of a single copy's Box (gymnasium.vector.utils.batch_space, without needing gymnasium)."""
    from .spaces import Box
    return Box(low=np.repeat(np.asarray(space.low)[None], n, axis=0), high=np.repeat(np.asarray(space.high)[None], n, axis=0))
