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
    truncations, infos)``.  ``actions`` is a numpy array or a torch CUDA tensor (outputs: torch tensors in HBM, the launch
    asynchronous on the current torch stream).  It is a ``gymnasium.vector.VectorEnv`` where gymnasium exists.

    ``copy`` (default True, like Gymnasium's own vector envs): what ``step`` returns is the caller's to keep, for good.
    numpy: fresh arrays, copied out of the handle's pinned host buffers; torch: fresh tensors every step (four
    allocations from torch's caching allocator, no synchronisation; the kernel writes into them directly).
    ``copy=False`` is the zero-copy contract: numpy views of the pinned buffers / one persistent set of torch tensors,
    overwritten in place by the next step and gone after ``close()`` -- for samplers that consume a step's results
    before taking the next.

    The adapter puts the handle into the one-agent I/O layout (``mjrl_set_io_layout``): the kernel reads the driven
    agent's action row straight from the caller's ``[num_envs, act_dim]`` array / tensor (no padded copy; the other
    agents act with 0) and writes only its observation row.  ``obs_dtype=np.float32`` makes the kernel store the
    observations as float -- what a policy network consumes, and half the bytes a numpy caller pulls over PCIe.  The env
    object's own array path (``step_batched`` with all agents' rows) is unavailable while the adapter holds it.

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

    def __init__(self, environment, agent: str | None = None, autoreset: str = "next_step", copy: bool = True,
                 obs_dtype=np.float64):
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
        if np.dtype(obs_dtype) not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise Exception("obs_dtype must be float64 or float32")
        self._f32 = np.dtype(obs_dtype) == np.dtype(np.float32)
        self._obs_dim = environment._handle.size("obs_dim")
        if getattr(environment, "_latent_dim", 0):
            raise Exception("BatchedVectorEnv: camera latents in the observation use the all-agent layout; step the env "
                            "through MuJoCoRL.step_batched")
        environment._handle.set_io_layout(self._k, self._f32)
        environment._io_layout = (self._k, self._f32, self.MODES[autoreset])       # (kept across a level switch on reset)
        environment._pinned = None                 # (views of the old layout)
        self._torch_act = None
        self._copy = bool(copy)
        self._outputs = None              # (copy=False: the one persistent set of torch output tensors)
        environment._handle.set_autoreset(self.MODES[autoreset])

    # -- helpers
    def _pinned_buffers(self):
        env = self.environment
        if env._pinned is None or env._pinned[0] is not env._handle or env._pinned[1].shape[-1] != self._act_dim:
            env._pinned = (env._handle,) + env._handle.host_buffers(self._act_dim)
            env._pinned[1][:] = 0.0
        return env._pinned

    # -- VectorEnv
    def reset(self, *, seed=None, options=None):
        env = self.environment
        env.reset_batched()
        obs = np.atleast_2d(env.get_observations(self.agent))
        full = np.zeros((self.num_envs, self._width), np.float32 if self._f32 else np.float64)
        full[:, :obs.shape[1]] = obs              # (slots of fused dynamics read 0 after a reset, like the array path's)
        return full, {}

    def step(self, actions):
        env = self.environment
        k = self._k
        if isinstance(actions, np.ndarray) or not hasattr(actions, "data_ptr"):
            _, p_act, obs, reward, term, trunc = self._pinned_buffers()
            act = np.asarray(actions, dtype=np.float64).reshape(self.num_envs, -1)
            p_act[:, :act.shape[1]] = act
            env._handle.step_pinned(self._act_dim, env.skip_frames)
            env.timestep += 1
            env._obs_cache = None
            obs, reward, term, trunc = obs[:, :self._width], reward[:, k], term[:, k].view(np.bool_), trunc[:, k].view(np.bool_)
            if self._copy:              # (the pinned buffers are the next step's too, and close() frees them)
                obs, reward, term, trunc = obs.copy(), reward.copy(), term.copy(), trunc.copy()
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
        if self._copy or self._outputs is None or self._outputs[0].device != actions.device:
            n, na, dev = self.num_envs, self._n_agent, actions.device
            fresh = (torch.empty((n, self._obs_dim), dtype=torch.float32 if self._f32 else torch.float64, device=dev),
                     torch.empty((n, na), dtype=torch.float64, device=dev),
                     torch.empty((n, na), dtype=torch.uint8, device=dev),
                     torch.empty((n, na), dtype=torch.uint8, device=dev))
            if not self._copy:
                self._outputs = fresh
        else:
            fresh = self._outputs
        o_obs, o_rew, o_term, o_trunc = fresh
        # (raw addresses go to the kernel: the action tensor is checked on every call, the outputs were made here)
        if not (actions.is_cuda and actions.device.index == env.device_id and actions.dtype == torch.float64
                and actions.is_contiguous() and actions.dim() == 2 and actions.shape[0] == self.num_envs
                and actions.shape[1] == self._act_dim):
            if actions.is_cuda and actions.dim() == 2 and actions.shape[0] == self.num_envs and actions.shape[1] <= self._act_dim:
                # another dtype / a narrower row: through a staging tensor of the layout's shape
                if self._torch_act is None or self._torch_act.device != actions.device:
                    self._torch_act = torch.zeros((self.num_envs, self._act_dim), dtype=torch.float64, device=actions.device)
                self._torch_act[:, :actions.shape[1]] = actions
                actions = self._torch_act
            else:
                raise Exception(f"BatchedVectorEnv.step: actions {tuple(actions.shape)} {actions.dtype} on {actions.device}; "
                                f"expected ({self.num_envs}, {self._act_dim}) float64 on cuda:{env.device_id}")
        stream = torch.cuda.current_stream(actions.device).cuda_stream
        if stream != env._stream:
            env.set_stream(stream)
        env._handle.step_device(actions.data_ptr(), self._act_dim, env.skip_frames, o_obs.data_ptr(), o_rew.data_ptr(),
                                o_term.data_ptr(), o_trunc.data_ptr())
        env.timestep += 1
        env._obs_cache = None
        # (flags as bool views of the kernel's bytes: no conversion kernel)
        obs, reward, term, trunc = o_obs[:, :self._width], o_rew[:, k], o_term[:, k].view(torch.bool), o_trunc[:, k].view(torch.bool)
        info = {}
        if self.autoreset == "same_step":
            done = term | trunc
            if bool(done.any()):              # (a host look at the flags: this mode's price)
                info = {"final_observation": obs[done].clone(), "_final_observation": done.clone()}
                fresh = torch.empty_like(o_obs)
                env._handle.reset_device(done.to(torch.uint8).contiguous().data_ptr(), fresh.data_ptr())
                obs = torch.where(done[:, None], fresh[:, :self._width], obs)
        return obs, reward, term, trunc, info

    def close(self, **kwargs):
        self._outputs = None
        self._torch_act = None
        self.environment.close()


def _batch_space(space, n):
    """The batched Box of a single copy's Box (gymnasium.vector.utils.batch_space, without needing gymnasium)."""
    from .spaces import Box
    return Box(low=np.repeat(np.asarray(space.low)[None], n, axis=0), high=np.repeat(np.asarray(space.high)[None], n, axis=0))
