"""``MuJoCoRL``: the reference's multi-agent env surface (MuJoCo_Gym/mujoco_rl.py) over the batched stepper.

Same ``config_dict`` keys and defaults (mujoco_rl.py:51-64), same ``reset()`` / ``step()`` return structure,
same environmentDynamics / rewardFunctions / doneFunctions plugin contract (4-tuple ``dynamic``), same call
order (dynamic-major, agent-minor; rewards after all dynamics; truncation before the counter moves).
Extra keys: ``numEnvs`` (default 1), ``deviceId`` (default 0), ``nconmax`` / ``njmax``, ``firstEnvId`` (global id of
copy 0 when the batch is a shard; on-device random choices are keyed on the global copy id), ``variantSeed``,
``fusedPlugins`` (default True).

With ``numEnvs == 1`` every value has the reference's shape, so the loops of
benchmarking/different_env_configs/fps_benchmark.py:33-41 run unchanged.  With more copies each per-agent
value gains a leading ``[numEnvs]`` axis and plugins are called once per (dynamic, agent) with batched views.
``step_batched`` is the array-in / array-out path without per-agent dicts; it keeps everything in HBM when
given torch CUDA tensors.
"""
from __future__ import annotations

import copy
import json
import os
import time

import numpy as np

from . import dynamics as fused_vocabulary
from .helper import update_deep
from .mujoco_parent import MuJoCoParent
from .spaces import Box

try:  # pragma: no cover - depends on the environment
    # the reference's class is a pettingzoo ParallelEnv (mujoco_rl.py:7, 18); so is this one where pettingzoo exists
    from pettingzoo import ParallelEnv as _ParallelBase
except Exception:
    class _ParallelBase:
        pass


class MuJoCoRL(_ParallelBase, MuJoCoParent):
    metadata = {"name": "mjrl_amd_v0", "render_modes": ["none"]}

    _pinned = None          # (handle, actions, obs, reward, term, trunc): the handle's pinned host buffers, step_batched
    _checked_buffers = None  # output buffers of step_batched that have passed _check_device_buffers, by address
    _act_need = None         # action slots step_batched needs per agent

    def __init__(self, config_dict: dict):
        self.agents = config_dict.get("agents", [])
        self.possible_agents = self.agents
        self.xml_paths = config_dict.get("xmlPath")
        self.info_jsons = config_dict.get("infoJson", None)
        self.render_mode = config_dict.get("renderMode", False)
        self.export_path = config_dict.get("exportPath")
        self.free_joint = config_dict.get("freeJoint", False)
        self.skip_frames = config_dict.get("skipFrames", 1)
        self.max_steps = config_dict.get("maxSteps", 1024)
        self.reward_functions = config_dict.get("rewardFunctions", [])
        self.done_functions = config_dict.get("doneFunctions", [])
        self.environment_dynamics = config_dict.get("environmentDynamics", [])
        self.agent_cameras = config_dict.get("agentCameras", False)
        sensor_resolution = config_dict.get("sensorResolution", (64, 64))
        # {"weights": {w1, b1, w2, b2, wd, bd}, "relu": True}: the encoder of vision/autoencoder.py:12-18; with
        # agentCameras every agent's observation then ends with the latent of its first camera's 64x64 image
        self.camera_encoder = config_dict.get("cameraEncoder")

        self.timestep = 0
        self._checked_buffers = {}
        self.start_time = time.time()
        self.action_routing = {"physical": [], "dynamic": {}}
        self.data_store = {agent: {} for agent in self.agents}

        MuJoCoParent.__init__(self, xml_paths=self.xml_paths, export_path=self.export_path, render=self.render_mode,
                              free_joint=self.free_joint, agent_cameras=self.agent_cameras,
                              sensor_resolution=sensor_resolution, n_env=config_dict.get("numEnvs", 1),
                              device_id=config_dict.get("deviceId", 0), nconmax=config_dict.get("nconmax"),
                              njmax=config_dict.get("njmax"), first_env_id=config_dict.get("firstEnvId", 0),
                              variant_seed=config_dict.get("variantSeed", 0),
                              shares_device=config_dict.get("sharesDevice", False))
        self._handle.set_max_steps(self.max_steps)
        self._load_info_json()

        self.environment_dynamics = [dynamic(self) for dynamic in self.environment_dynamics]
        self._check_dynamics(self.environment_dynamics)
        self._check_reward_functions(self.reward_functions)
        self._check_done_functions(self.done_functions)

        self._observation_space = self._create_observation_space()
        self._first_observation_space = self._observation_space[list(self._observation_space.keys())[0]]
        self._action_space = self._create_action_space()
        self._first_action_space = self._action_space[list(self._action_space.keys())[0]]
        self._upload_tables(self.agents)
        self._fused_allowed = config_dict.get("fusedPlugins", True)
        self._setup_fused_program(self._fused_allowed)
        self._setup_camera_encoder()

    def _setup_camera_encoder(self):
        """Camera latents as the tail of the observation (mjrl_set_camera_obs): what a vision policy of the reference
        does by hand -- ``get_camera_data(agent)`` (mujoco_parent.py:540-555), scale by 1/255, encode
        (vision/autoencoder.py:12-18) -- runs behind every step launch on the device."""
        self._latent_dim = 0
        if not self.camera_encoder:
            return
        if not self.agent_cameras:
            raise Exception("cameraEncoder needs agentCameras: True")
        self._handle.encoder_load(self.camera_encoder["weights"], relu=self.camera_encoder.get("relu", True))
        names = self._compiled.names["camera"]
        cams = [names.index(self.rgb_sensors[a][0]) if self.rgb_sensors.get(a) else -1 for a in self.agents]
        self._handle.set_camera_obs(cams)
        self._latent_dim = self._handle.latent_dim
        low = 0.0 if self.camera_encoder.get("relu", True) else -np.inf
        for agent, space in self._observation_space.items():
            self._observation_space[agent] = Box(low=np.concatenate([space.low, np.full(self._latent_dim, low)]),
                                                 high=np.concatenate([space.high, np.full(self._latent_dim, np.inf)]))
        self._first_observation_space = self._observation_space[list(self._observation_space.keys())[0]]

    def _camera_latents(self, agent):
        """The latent slots of ``agent``'s row in the last step's observation buffer (host plugin path)."""
        k = self._table_agents.index(agent)
        start = self._obs_len[agent] + (self._program.n_extra if self._program is not None else 0)
        if self._obs_cache is None:
            return self._squeeze(np.zeros((self.n_env, self._latent_dim)))
        return self._squeeze(self._obs_cache[:, k, start:start + self._latent_dim].copy())

    def _after_init_environment(self):
        """A multi-level ``reset()`` re-created the device state for another level: the new handle gets everything the
        old one had -- tables, truncation horizon, the fused plugin program rebuilt against the new level's body / geom
        ids (or the query cache for host plugins).  The reference re-creates model and data and keeps its Python plugin
        loop (mujoco_parent.py:351-356)."""
        MuJoCoParent._after_init_environment(self)
        self._handle.set_max_steps(self.max_steps)
        if getattr(self, "_io_layout", None):            # (a vector-env adapter's one-agent layout and autoreset mode)
            self._handle.set_io_layout(*self._io_layout[:2])
            self._handle.set_autoreset(self._io_layout[2])
        if hasattr(self, "_fused_allowed"):
            self._setup_fused_program(self._fused_allowed)
            if self.camera_encoder:
                self._handle.encoder_load(self.camera_encoder["weights"], relu=self.camera_encoder.get("relu", True))
                names = self._compiled.names["camera"]
                self._handle.set_camera_obs([names.index(self.rgb_sensors[a][0]) if self.rgb_sensors.get(a) else -1
                                             for a in self.agents])

    def _setup_fused_program(self, allowed: bool):
        """When every configured plugin belongs to the device vocabulary (dynamics.py) the plugin loop runs inside
        the step kernel; otherwise (or with ``fusedPlugins: False``) it stays the host loop below."""
        self._program = fused_vocabulary.build_program(self) if allowed else None
        self._act_need = None            # (what step_batched checks the action tensors against)
        if self._program is None:
            # host plugins read body / geom frames and contacts after every step: have the kernel keep them
            if self.environment_dynamics or self.reward_functions or self.done_functions:
                self._handle.set_query_cache(True)
            return
        names = self._compiled.names["body"]
        for agent in self.agents:
            if agent not in names:
                raise Exception(f"agent {agent} is not a body of the level")
        if self._program.tags:
            self._handle.set_tag_tables([self.tag_refs(tag) for tag in self._program.tags])
        pi, pf = self._program.arrays()
        self._handle.set_program(pi, pf, len(self._program.slots), self._program.n_extra,
                                 [names.index(agent) for agent in self.agents])

    @property
    def device_store(self):
        """The fused program's per-agent data store: ``{agent: {key: array[numEnvs]}}`` (NaN = key absent)."""
        if self._program is None or not self._program.slots:
            return {agent: {} for agent in self.agents}
        raw = self._handle.get_field("store")
        return {agent: {key: raw[:, k, slot] for key, slot in self._program.slots.items()}
                for k, agent in enumerate(self.agents)}

    # ------------------------------------------------------------------ construction helpers
    def _load_info_json(self):
        """mujoco_rl.py:93-112."""
        path = None
        if isinstance(self.info_jsons, list):
            if len(self.info_jsons) != len(self.xml_paths):
                raise Exception("Length mismatch between info_json list and xml_paths list")
            stem = os.path.split(self.xml_path)[1].split(".")[0] + ".json"
            path = [candidate for candidate in self.info_jsons if stem in candidate][0]
        elif isinstance(self.info_jsons, str):
            path = self.info_jsons
        if path is None:
            self.info_json, self.info_name_list = None, []
            return
        with open(path) as fh:
            self.info_json = json.load(fh)
        self.info_name_list = list(self.info_json["environment"]["objects"].keys())

    def _check_dynamics(self, dynamics):
        """Construction-time validation (mujoco_rl.py:114-143): one trial call per class with the lower action
        bound; its side effects on ``data_store`` are kept, as in the reference."""
        for instance in dynamics:
            trial = instance.action_space["low"]
            reward, observations, done, info = instance.dynamic(self.agents[0], trial)
            low, high = instance.observation_space["low"], instance.observation_space["high"]
            obs = np.asarray(observations)
            width = obs.shape[-1] if obs.ndim else 1
            if len(low) != width:
                raise Exception(f"Observation, the second return variable of dynamic function, must match length"
                                f" of lower bound of observation space of {instance}")
            if not np.all(np.asarray(low) <= obs):
                raise Exception(f"Observation, the second return variable of dynamic function, exceeds the lower bound"
                                f" on at least one axis of the observation space of {instance}")
            if len(high) != width:
                raise Exception(f"Observation, the second return variable of dynamic function, must match length of"
                                f" upper bound of observation space of {instance}")
            if not np.all(np.asarray(high) >= obs):
                raise Exception(f"Observation, the second return variable of dynamic function, exceeds the upper bound"
                                f" on at least one axis of the observation space of {instance}")
            if not (isinstance(reward, (float, int, np.floating, np.integer)) or
                    (self.n_env > 1 and isinstance(reward, np.ndarray))):
                raise Exception(f"Reward, the first return variable of dynamic function of {instance}, must be a float")

    def _check_done_functions(self, done_functions):
        for fn in done_functions:
            done = fn(self, self.agents[0])
            if not (isinstance(done, (int, bool, np.bool_, np.integer)) or (self.n_env > 1 and isinstance(done, np.ndarray))):
                raise Exception(f"Done, the first return variable of {fn}, must be a boolean")

    def _check_reward_functions(self, reward_functions):
        for fn in reward_functions:
            reward = fn(self, self.agents[0])
            if not (isinstance(reward, (float, int, np.floating, np.integer)) or
                    (self.n_env > 1 and isinstance(reward, np.ndarray))):
                raise Exception(f"Reward, the second return variable of {fn}, must be a float")

    def _create_action_space(self) -> dict:
        """Physical actuators first, then one slice per dynamics class in list order (mujoco_rl.py:171-193)."""
        spaces = {}
        for agent in self.agents:
            space = self.get_action_space_mujoco(agent)
            self.action_routing["physical"] = [0, len(space["low"])]
            for dynamic in self.environment_dynamics:
                extra = dynamic.action_space
                start = len(space["low"])
                self.action_routing["dynamic"][dynamic.__class__.__name__] = [start, start + len(extra["low"])]
                space["low"] += extra["low"]
                space["high"] += extra["high"]
            spaces[agent] = Box(low=np.array(space["low"]), high=np.array(space["high"]))
        return spaces

    def _create_observation_space(self) -> dict:
        spaces = {}
        for agent in self.agents:
            space = self.get_observation_space_mujoco(agent)
            for dynamic in self.environment_dynamics:
                space["low"] += dynamic.observation_space["low"]
                space["high"] += dynamic.observation_space["high"]
            spaces[agent] = Box(low=np.array(space["low"]), high=np.array(space["high"]))
        return spaces

    # ------------------------------------------------------------------ plugin loop
    def _apply_dynamics(self, action, observations, rewards, terminations, infos):
        """Dynamic-major, agent-minor, strictly sequential: each call sees the data_store writes of the calls
        before it (mujoco_rl.py:215-241)."""
        for dynamic in self.environment_dynamics:
            name = dynamic.__class__.__name__
            lo, hi = self.action_routing["dynamic"][name]
            for agent in self.agents:
                act = np.asarray(action[agent])
                reward, obs, done, info = dynamic.dynamic(agent, act[..., lo:hi])
                obs = np.asarray(obs, dtype=np.float64)
                if self.n_env > 1 and obs.ndim == 1:
                    obs = np.broadcast_to(obs, (self.n_env, obs.shape[0]))
                observations[agent] = np.concatenate((observations[agent], obs), axis=-1)
                rewards[agent] = rewards[agent] + reward
                terminations[agent] = np.logical_or(terminations[agent], done) if self.n_env > 1 else any([terminations[agent], done])
                infos[agent][name] = info
        return observations, rewards, terminations, infos

    def _blank(self, value):
        return value if self.n_env == 1 else np.full(self.n_env, value)

    def _step_fused(self, action: dict):
        """step() when the plugin loop runs on the device: one launch, then only the dict packaging."""
        width = self._first_action_space.shape[0]
        arr = np.zeros((self.n_env, len(self.agents), width), np.float64)
        for k, agent in enumerate(self.agents):
            act = np.asarray(action[agent], dtype=np.float64).reshape(self.n_env, -1)
            if act.shape[-1] < width:
                raise Exception(f"The number of actions for agent {agent} is not correct.")
            arr[:, k] = act[:, :width]
        n_agent, obs_dim = len(self.agents), self._handle.size("obs_dim")
        obs = np.zeros((self.n_env, n_agent, obs_dim))
        rew = np.zeros((self.n_env, n_agent))
        term = np.zeros((self.n_env, n_agent), np.uint8)
        trunc = np.zeros((self.n_env, n_agent), np.uint8)
        self._handle.step_host(arr, self.skip_frames, obs, rew, term, trunc)
        self.frame += self.skip_frames
        self._obs_cache = None
        extra = self._program.n_extra + self._latent_dim
        squeeze = (lambda x: x[0]) if self.n_env == 1 else (lambda x: x)
        observations = {a: squeeze(obs[:, k, :self._obs_len[a] + extra].copy()) for k, a in enumerate(self.agents)}
        rewards = {a: squeeze(rew[:, k].copy()) for k, a in enumerate(self.agents)}
        terminations = {a: squeeze(term[:, k].astype(bool)) for k, a in enumerate(self.agents)}
        if self.n_env == 1:
            rewards = {a: float(v) for a, v in rewards.items()}
            terminations = {a: bool(v) for a, v in terminations.items()}
        infos = {a: {d.__class__.__name__: {} for d in self.environment_dynamics} for a in self.agents}
        # the kernel's own flags: evaluated per copy against the copy's own step counter (mujoco_rl.py:406-417 before the
        # counter moves), so that copies reset on their own (reset_batched(mask), in-launch resets) are told apart --
        # the single host counter below is copy 0's
        flags = trunc[:, 0].astype(bool)
        truncations = {a: (bool(flags[0]) if self.n_env == 1 else flags.copy()) for a in self.agents}
        truncations["__all__"] = bool(flags[0]) if self.n_env == 1 else flags.copy()
        if len(self.done_functions) != 0:
            if self.n_env == 1:
                terminations["__all__"] = any(terminations.values())
            else:
                terminations["__all__"] = np.logical_or.reduce([terminations[a] for a in self.agents])
        self.timestep += 1
        return observations, rewards, terminations, truncations, infos

    def step(self, action: dict):
        """mujoco_rl.py:243-289."""
        if self._program is not None:
            return self._step_fused(action)
        lo, hi = self.action_routing["physical"]
        physical = {agent: np.asarray(action[agent])[..., lo:hi] for agent in action.keys()}
        self.apply_action(physical, skip_frames=self.skip_frames)

        observations = {agent: self.get_observations(agent) for agent in self.agents}
        rewards = {agent: self._blank(0) for agent in self.agents}
        terminations = {agent: self._blank(False) for agent in self.agents}
        infos = {agent: {} for agent in self.agents}
        observations, rewards, terminations, infos = self._apply_dynamics(action, observations, rewards, terminations, infos)
        if self._latent_dim:
            observations = {agent: np.concatenate((observations[agent], self._camera_latents(agent)), axis=-1) for agent in self.agents}

        for reward_fn in self.reward_functions:
            rewards = {agent: rewards[agent] + reward_fn(self, agent) for agent in self.agents}

        truncations = self._check_truncations()

        if len(self.done_functions) != 0:
            for done_fn in self.done_functions:
                if self.n_env == 1:
                    terminations = {agent: any([terminations[agent], done_fn(self, agent)]) for agent in self.agents}
                    terminations["__all__"] = any(terminations.values())
                    if terminations["__all__"]:
                        break
                else:
                    terminations = {agent: np.logical_or(terminations[agent], done_fn(self, agent)) for agent in self.agents}
                    terminations["__all__"] = np.logical_or.reduce([terminations[a] for a in self.agents])
                    if np.all(terminations["__all__"]):
                        break

        self.timestep += 1
        return observations, rewards, terminations, truncations, infos

    def reset(self, *, seed: int = None, options=None):
        """mujoco_rl.py:291-331: reset the physics, clear the data store, run the dynamics once with a random
        action so the observation has its full width, then discard what that pass wrote to the data store."""
        MuJoCoParent.reset(self)
        if isinstance(self.info_jsons, list):
            self._load_info_json()
        self.data_store = {agent: {} for agent in self.agents}

        observations = {agent: self.get_observations(agent) for agent in self.agents}
        sample = self._first_action_space.sample
        action = {agent: (sample() if self.n_env == 1 else np.stack([sample() for _ in range(self.n_env)]))
                  for agent in self.agents}
        rewards = {agent: self._blank(0) for agent in self.agents}
        terminations = {agent: self._blank(False) for agent in self.agents}
        infos = {agent: {} for agent in self.agents}
        copies = [copy.deepcopy(self.data_store) for _ in range(len(self.environment_dynamics))]
        original = copy.deepcopy(self.data_store)
        observations, rewards, terminations, infos = self._apply_dynamics(action, observations, rewards, terminations, infos)
        if self._latent_dim:        # (the first images are encoded by the first step; the reset observation carries zeros)
            observations = {agent: np.concatenate((observations[agent], self._camera_latents(agent)), axis=-1) for agent in self.agents}
        self.data_store = original
        for stored in copies:
            self.data_store = update_deep(self.data_store, stored)
        self.timestep = 0
        return observations, infos

    def _check_truncations(self) -> dict:
        """mujoco_rl.py:406-417 (evaluated before ``timestep`` moves)."""
        flag = self.timestep >= self.max_steps
        truncations = {agent: self._blank(flag) for agent in self.agents}
        truncations["__all__"] = self._blank(flag)
        return truncations

    # ------------------------------------------------------------------ spaces / info
    def action_space(self, agent: str):
        return self._action_space[agent]

    def observation_space(self, agent: str):
        return self._observation_space[agent]

    def tagged_names(self, tag: str) -> list:
        """Names of the objects that carry ``tag``, in the order ``filter_by_tag`` visits them: the environment's objects,
        then every area's (mujoco_rl.py:364-377)."""
        found = []
        groups = [self.info_json["environment"]["objects"]]
        groups += [self.info_json["areas"][area]["objects"] for area in self.info_json["areas"]]
        for objects in groups:
            for name in objects:
                if "tags" in objects[name].keys() and objects[name]["tags"] is not None and tag in objects[name]["tags"]:
                    found.append(name)
        return found

    def filter_by_tag(self, tag: str) -> list:
        """mujoco_rl.py:355-378."""
        return [self.get_data(name) for name in self.tagged_names(tag)]

    def tag_refs(self, tag: str) -> list:
        """The device form of a tag (mjrl_set_tag_tables): ``(kind, id)`` per tagged object, kind 0 = body (its position is
        xipos), 1 = geom (xpos) -- resolved like ``get_data`` resolves a name (body first, mujoco_parent.py:404-426)."""
        names = self._compiled.names
        refs = []
        for name in self.tagged_names(tag):
            folded = getattr(self._compiled, "folded_bodies", {})
            if name in folded and folded[name]["geoms"]:
                # a static body folded into the world: its (single, centred) geom stands for it -- geom xpos = body xipos
                refs.append((1, folded[name]["geoms"][0]))
            elif name in names["body"]:
                refs.append((0, names["body"].index(name)))
            elif name in names["geom"]:
                refs.append((1, names["geom"].index(name)))
            else:
                raise KeyError(f"Invalid name '{name}'")
        return refs

    def get_data(self, name: str) -> dict:
        """mujoco_rl.py:380-395: physics record merged with the info-JSON attributes."""
        data = MuJoCoParent.get_data(self, name)
        if name in self.info_name_list:
            for key, value in self.info_json["environment"]["objects"][name].items():
                if key not in ("position", "orientation", "mass"):
                    data[key] = value
        return data

    def close(self):
        """Release the device state (and the views of the handle's pinned host buffers).  Defined here, not only on the
        parent: where pettingzoo is installed ``ParallelEnv.close`` -- a no-op -- comes first in the method resolution
        order and would shadow it (the reference's parent has no close at all; this port's owns HBM)."""
        self._pinned = None
        self._checked_buffers = {}
        MuJoCoParent.close(self)

    # ------------------------------------------------------------------ array path
    def reset_batched(self, mask=None):
        """Reset every copy (or the copies flagged in ``mask``) without building per-agent dicts; asynchronous on
        the handle's stream.  The array-path counterpart of ``reset()`` for runs without host plugins."""
        self._handle.reset(mask)
        self.timestep = 0 if mask is None else self.timestep
        self._episode = None
        self._obs_cache = None

    def _check_device_buffers(self, actions, obs, reward, term, trunc):
        """The kernel gets raw addresses: a tensor of the wrong shape, type, layout or device would be an out-of-bounds
        device access.  The output buffers are checked once per set (``_checked_buffers``), the actions on every call
        (a sampler hands over a different action tensor every step; the test is a handful of attribute reads)."""
        import torch
        n_agent = len(self.agents)
        if self._act_need is None:
            routed = max((len(self.agents_action_index[a]) for a in self.agents), default=0)
            self._act_need = max(routed, self._first_action_space.shape[0] if self._program is not None else 0, 1)
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.device.index == self.device_id
                and actions.dtype == torch.float64 and actions.is_contiguous() and actions.dim() == 3
                and actions.shape[0] == self.n_env and actions.shape[1] == n_agent and actions.shape[2] >= self._act_need):
            what = tuple(actions.shape) if isinstance(actions, torch.Tensor) else type(actions).__name__
            raise Exception(f"step_batched: actions ({what}, {getattr(actions, 'dtype', None)}, {getattr(actions, 'device', None)}) "
                            f"must be a contiguous float64 tensor of shape ({self.n_env}, {n_agent}, >= {self._act_need}) on "
                            f"cuda:{self.device_id}")
        key = (self._handle, obs.data_ptr() if isinstance(obs, torch.Tensor) else None,
               reward.data_ptr() if isinstance(reward, torch.Tensor) else None,
               term.data_ptr() if isinstance(term, torch.Tensor) else None,
               trunc.data_ptr() if isinstance(trunc, torch.Tensor) else None)
        if key in self._checked_buffers:
            return
        obs_dim = self._handle.size("obs_dim")
        expect = {"obs": (obs, torch.float64, (self.n_env, n_agent, obs_dim)),
                  "reward": (reward, torch.float64, (self.n_env, n_agent)), "term": (term, torch.uint8, (self.n_env, n_agent)),
                  "trunc": (trunc, torch.uint8, (self.n_env, n_agent))}
        for name, (tensor, dtype, shape) in expect.items():
            if not isinstance(tensor, torch.Tensor):
                raise Exception(f"step_batched: {name} must be a torch tensor when the actions are one")
            if not tensor.is_cuda or tensor.device.index != self.device_id:
                raise Exception(f"step_batched: {name} lives on {tensor.device}, the env batch on cuda:{self.device_id}")
            if tensor.dtype != dtype or not tensor.is_contiguous():
                raise Exception(f"step_batched: {name} must be a contiguous {dtype} tensor")
            if tuple(tensor.shape) != shape:
                raise Exception(f"step_batched: {name} has shape {tuple(tensor.shape)}, expected {shape}")
        if len(self._checked_buffers) > 64:
            self._checked_buffers.clear()
        # (the tensors are kept alive with their key: an address must not pass for another tensor's later)
        self._checked_buffers[key] = (obs, reward, term, trunc)

    def step_batched(self, actions, obs=None, reward=None, term=None, trunc=None):
        """One step of every copy without per-agent dicts and without host plugins.

        ``actions``: ``[numEnvs, n_agent, act_dim]`` float64, numpy array or torch CUDA tensor.  With torch
        tensors nothing leaves HBM and the launch is asynchronous on the current torch stream; outputs are
        torch tensors (pass preallocated ones to avoid allocations).  With a numpy array and no output arrays the
        step runs on the handle's pinned host buffers (``mjrl_step_pinned``: the kernel reads the actions and writes
        the results over PCIe itself, no copies) and the returned arrays are views of those buffers -- the next step
        overwrites them, copy what has to survive it; pass output arrays to get copies (``mjrl_step_host``).
        Returns ``(obs, reward, term, trunc)``.
        """
        if self._program is None and (self.environment_dynamics or self.reward_functions or self.done_functions):
            raise Exception("step_batched runs no host plugins; use step() or the fused vocabulary (dynamics.py)")
        n_agent, obs_dim = len(self.agents), self._handle.size("obs_dim")
        if isinstance(actions, np.ndarray):
            if obs is None and reward is None and term is None and trunc is None:
                act_dim = int(actions.shape[-1])
                if self._pinned is None or self._pinned[0] is not self._handle or self._pinned[1].shape[-1] != act_dim:
                    self._pinned = (self._handle,) + self._handle.host_buffers(act_dim)
                _, p_act, obs, reward, term, trunc = self._pinned
                np.copyto(p_act, actions.reshape(p_act.shape))
                self._handle.step_pinned(act_dim, self.skip_frames)
            else:
                actions = np.ascontiguousarray(actions, dtype=np.float64)
                obs = np.zeros((self.n_env, n_agent, obs_dim)) if obs is None else obs
                reward = np.zeros((self.n_env, n_agent)) if reward is None else reward
                term = np.zeros((self.n_env, n_agent), np.uint8) if term is None else term
                trunc = np.zeros((self.n_env, n_agent), np.uint8) if trunc is None else trunc
                self._handle.step_host(actions, self.skip_frames, obs, reward, term, trunc)
        else:
            import torch
            dev = actions.device
            obs = torch.empty((self.n_env, n_agent, obs_dim), dtype=torch.float64, device=dev) if obs is None else obs
            reward = torch.empty((self.n_env, n_agent), dtype=torch.float64, device=dev) if reward is None else reward
            term = torch.empty((self.n_env, n_agent), dtype=torch.uint8, device=dev) if term is None else term
            trunc = torch.empty((self.n_env, n_agent), dtype=torch.uint8, device=dev) if trunc is None else trunc
            self._check_device_buffers(actions, obs, reward, term, trunc)
            stream = torch.cuda.current_stream(dev).cuda_stream
            if stream != self._stream:
                self.set_stream(stream)
            self._handle.step_device(actions.data_ptr(), actions.shape[-1], self.skip_frames, obs.data_ptr(),
                                     reward.data_ptr(), term.data_ptr(), trunc.data_ptr())
        self.timestep += 1
        self._obs_cache = None
        return obs, reward, term, trunc
