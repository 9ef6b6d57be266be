"""Sim host of the batched stepper -- the counterpart of the reference's ``MuJoCoParent``
(MuJoCo_Gym/mujoco_parent.py).

Same constructor arguments, same table builders and query helpers, but ``n_env`` copies of the level live
in HBM behind the C-ABI of libmjrl_hip.so and one kernel launch advances all of them.  Interactive GLFW
rendering (mujoco_parent.py:99-105, 577-616) is out of scope; agent cameras are served by the ray-cast
render entry when it is available.

Index tables are built by walking the xmltodict-shaped level dict exactly like the reference does, so the
gather / scatter indices are bit-identical (golden values: tests/golden/index_tables.json).
"""
from __future__ import annotations

import math
import random
import warnings

import numpy as np

from . import _capi, blob, mjcf, xmldict
from .helper import mat2euler_scipy
from .sensor import create_sensor_observation_space, process_sensors

_SITE_SENSOR_KEYS = ("rangefinder", "touch", "accelerometer")


class _Named:
    """Attribute bag returned by ``data.body(name)`` / ``model.geom(name)`` style accessors."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class ModelView:
    """The slice of ``MjModel`` that plugins and the table builders read."""

    def __init__(self, compiled: mjcf.Model):
        self._m = compiled
        self.nq, self.nv, self.nu, self.nbody, self.ngeom = compiled.nq, compiled.nv, compiled.nu, compiled.nbody, compiled.ngeom
        self.jnt_qposadr, self.jnt_dofadr, self.jnt_type = compiled.jnt_qposadr, compiled.jnt_dofadr, compiled.jnt_type
        self.opt = _Named(timestep=compiled.timestep, gravity=compiled.gravity)

    def joint(self, name):
        j = self._m.name2id("joint", name)
        return _Named(id=j, name=name, qposadr=self._m.jnt_qposadr[j:j + 1], dofadr=self._m.jnt_dofadr[j:j + 1],
                      type=self._m.jnt_type[j:j + 1])

    def body(self, name):
        b = self._m.name2id("body", name)
        return _Named(id=b, name=name, mass=self._m.body_mass[b:b + 1])

    def geom(self, name):
        g = self._m.name2id("geom", name)
        return _Named(id=g, name=name, rgba=self._m.geom_rgba[g], type=self._m.geom_type[g:g + 1],
                      size=self._m.geom_size[g])

    def camera(self, name):
        return _Named(id=self._m.name2id("camera", name), name=name)

    def sensor(self, name):
        s = self._m.name2id("sensor", name)
        return _Named(id=s, name=name, dim=int(self._m.sensor_dim[s]), adr=int(self._m.sensor_adr[s]))


class DataView:
    """The slice of ``MjData`` that plugins read.  Values are fetched from HBM on access; with one env copy
    the leading batch dimension is dropped so reference plugins work unchanged."""

    def __init__(self, parent):
        self._p = parent

    def _squeeze(self, arr):
        return arr[0] if self._p.n_env == 1 else arr

    @property
    def qpos(self): return self._squeeze(self._p._handle.get_field("qpos"))
    @property
    def qvel(self): return self._squeeze(self._p._handle.get_field("qvel"))
    @property
    def ctrl(self): return self._squeeze(self._p._handle.get_field("ctrl"))
    @property
    def sensordata(self): return self._squeeze(self._p._handle.get_field("sensordata"))
    @property
    def time(self): return self._squeeze(self._p._handle.get_field("timestep")) * self._p._compiled.timestep * max(self._p._skip_frames_hint, 1)

    def sensor(self, name):
        view = self._p.model.sensor(name)
        data = self._p._handle.get_field("sensordata")[:, view.adr:view.adr + view.dim]
        return _Named(id=view.id, name=name, data=self._squeeze(data))

    def body(self, name):
        folded = getattr(self._p._compiled, "folded_bodies", {})
        if name in folded:
            # a static body the model compiler folded into the world (levels with more bodies than a wavefront has lanes,
            # mjcf.fuse_static): it does not move, its frame is a constant of the model
            rec, n = folded[name], self._p.n_env
            rows = lambda v: self._squeeze(np.tile(np.asarray(v, np.float64), (n, 1)))
            xipos = self._p._folded_xipos(name)
            return _Named(id=-1, name=name, xpos=rows(rec["pos"]), xipos=rows(xipos), xquat=rows(rec["quat"]),
                          xmat=rows(mjcf.quat_to_mat(rec["quat"]).reshape(9)))
        b = self._p._compiled.name2id("body", name)
        q = self._p._handle.query
        xquat = q("xquat")[:, b]
        xmat = np.stack([mjcf.quat_to_mat(qq).reshape(9) for qq in xquat])
        return _Named(id=b, name=name, xpos=self._squeeze(q("xpos")[:, b]), xipos=self._squeeze(q("xipos")[:, b]),
                      xquat=self._squeeze(xquat), xmat=self._squeeze(xmat))

    def geom(self, name):
        g = self._p._compiled.name2id("geom", name)
        q = self._p._handle.query
        return _Named(id=g, name=name, xpos=self._squeeze(q("geom_xpos")[:, g]), xmat=self._squeeze(q("geom_xmat")[:, g]))

    @property
    def ncon(self):
        n = self._p._handle.query("ncon")[:, 0].astype(int)
        return int(n[0]) if self._p.n_env == 1 else n

    @property
    def contact(self):
        pairs = self._p._handle.query("contact_geom").astype(int)
        if self._p.n_env == 1:
            return [_Named(geom1=int(a), geom2=int(b)) for a, b in pairs[0]]
        return pairs


class MuJoCoParent:
    def __init__(self, xml_paths, export_path: str = None, render: bool = False, free_joint: bool = False,
                 agent_cameras: bool = False, sensor_resolution=(64, 64), n_env: int = 1, device_id: int = 0,
                 nconmax: int = None, njmax: int = None, first_env_id: int = 0, variant_seed: int = 0,
                 shares_device: bool = False):
        self.xml_paths = xml_paths
        # (other env objects step on this GPU at the same time: a small batch must not pick the step kernel's build that
        # assumes it has every SIMD to itself, _capi.Handle(few=False))
        self._shares_device = bool(shares_device)
        self.export_path = export_path
        self.render = render
        self.free_joint = free_joint
        self.agent_cameras = agent_cameras
        self.sensor_resolution = sensor_resolution
        self.n_env = int(n_env)
        self.device_id = int(device_id)
        self._nconmax, self._njmax = nconmax, njmax
        self.first_env_id, self._variant_seed = int(first_env_id), int(variant_seed)
        self._variants = None
        self.rgb_sensors = {}
        self.frame = 0
        self._skip_frames_hint = 1
        if render:
            raise Exception("renderMode needs an interactive GLFW window, which the batched stepper does not provide")
        if isinstance(xml_paths, str):
            self.xml_path = xml_paths
        elif isinstance(xml_paths, list):
            self.xml_path = random.choice(xml_paths)
        else:
            raise Exception("xmlPath must be a path or a list of paths")
        with open(self.xml_path, "r") as fh:
            self.xml_dict = xmldict.parse(fh.read())
        self._handle = None
        self._stream = None
        self._pinned = None           # (handle, actions, obs, reward, term, trunc): the handle's pinned host buffers
        self._detect_variants()
        self._init_environment()
        self.agents_action_index = {}
        self.agents_observation_index = {}
        self.agent_observations_id = []
        self._obs_cache = None

    # ------------------------------------------------------------------ model / device state
    def _detect_variants(self):
        """An ``xmlPath`` list whose levels are one model in different colours (Testing/levels/Model2-10.xml differ in the
        boxes' rgba only) does not need a new model per reset (mujoco_parent.py:351-356): the batch keeps ONE model and
        every copy draws its own variant at each of its resets (mjrl_set_variants).  Levels that differ in structure keep
        the reference's behaviour: reset() switches the whole batch to one randomly chosen level."""
        if not isinstance(self.xml_paths, list) or len(set(self.xml_paths)) < 2:
            return
        paths = list(dict.fromkeys(self.xml_paths))
        models = [mjcf.compile_mjcf(p, nconmax=self._nconmax, njmax=self._njmax) for p in paths]
        blobs = [blob.pack(m) for m in models]
        if all(blob.same_physics(blobs[0], b) for b in blobs[1:]):
            self._variants = {"paths": paths, "rgba": np.stack([np.asarray(m.geom_rgba, np.float64).reshape(-1, 4) for m in models])}
            self.xml_path = paths[0]
            with open(self.xml_path, "r") as fh:
                self.xml_dict = xmldict.parse(fh.read())

    def _init_environment(self):
        """Counterpart of mujoco_parent.py:119-137: compile the level and create the device copies."""
        self._compiled = mjcf.compile_mjcf(self.xml_path, nconmax=self._nconmax, njmax=self._njmax)
        self._blob = blob.pack(self._compiled)
        if self._handle is not None:
            self._handle.close()
        self._handle = _capi.Handle(self._blob, self.n_env, self.device_id, few=False if self._shares_device else None)
        if self._stream is not None:
            self._handle.set_stream(self._stream)
        if self.first_env_id:
            self._handle.set_env_base(self.first_env_id)
        if self._variants is not None:
            self._handle.set_variants(self._variants["rgba"], self._variant_seed)
        # images show MjData's frames as the last forward pass left them (mjv_updateScene, mujoco_parent.py:533): the
        # step kernel keeps them for the ray caster from the first step on
        self._scene_cache = bool(self.agent_cameras) and len(self._compiled.names["camera"]) > 0
        if self._scene_cache:
            self._handle.set_scene_cache(True)
        self.model = ModelView(self._compiled)
        self.data = DataView(self)
        self._episode = None

    @property
    def episode(self) -> np.ndarray:
        """Resets so far, per copy (part of the key of the plugins' random draws, dynamics.py); fetched from the device
        on first use after a reset."""
        if self._episode is None:
            self._episode = self._handle.get_field("episode")
        return self._episode

    def set_stream(self, hip_stream):
        """Launch on a caller-provided HIP stream (integer handle; None = the library's own).  Kept across the handle
        re-creations of a multi-level ``reset()``."""
        self._stream = hip_stream
        self._handle.set_stream(hip_stream)

    def _after_init_environment(self):
        """Hook: called when ``reset()`` has switched to another level of an ``xmlPath`` list and re-created the device
        state (mujoco_parent.py:351-356).  Whatever was configured on the old handle has to be configured again."""
        if getattr(self, "_table_agents", None):
            self._upload_tables(self._table_agents)

    @classmethod
    def tables_only(cls, xml_path: str, free_joint: bool = False, agent_cameras: bool = False):
        """Instance that can build index tables and spaces but owns no device state (no stepping): used to
        check the gather / scatter tables on a machine without a GPU."""
        self = object.__new__(cls)
        self.xml_paths = self.xml_path = xml_path
        self.free_joint, self.agent_cameras, self.render = free_joint, agent_cameras, False
        self.n_env, self.rgb_sensors, self._handle = 1, {}, None
        with open(xml_path, "r") as fh:
            self.xml_dict = xmldict.parse(fh.read())
        self._compiled = mjcf.compile_mjcf(xml_path)
        self.model = ModelView(self._compiled)
        self.agents_action_index, self.agents_observation_index = {}, {}
        return self

    def close(self):
        self._pinned = None
        if self._handle is not None:
            self._handle.close()
            self._handle = None

    # ------------------------------------------------------------------ table builders (init time)
    def _process_sensor(self, sensor, indices, key):
        current = self.model.sensor(sensor["@name"])
        indices[current.id] = {"name": sensor["@name"], "data": [0.0] * current.dim}
        if key in _SITE_SENSOR_KEYS:
            indices[current.id].update(site=sensor["@site"], type=key, cutoff=sensor["@cutoff"])
        if key == "framexaxis":
            # the reference files these under the type name "frameyaxis" (mujoco_parent.py:158-160)
            indices[current.id].update(site=sensor["@objname"], type="frameyaxis")

    def _create_sensor_index_dict(self, sensor_dict):
        indices = {}
        for block in sensor_dict:
            if block is None:
                continue
            for key, entry in block.items():
                if isinstance(entry, list):
                    for sensor in entry:
                        self._process_sensor(sensor, indices, key)
                elif isinstance(entry, dict):
                    self._process_sensor(entry, indices, key)
        return indices

    def retrieve_mujoco_all_indices(self, joint_dicts):
        """qpos / qvel index lists of the named joints: free -> 7 / 6, anything else -> 1 / 1
        (mujoco_parent.py:185-231)."""
        qpos_indices, qvel_indices = [], []
        for joint in joint_dicts:
            name = joint.get("@name")
            if not name:
                continue
            jid = self._compiled.name2id("joint", name)
            free = self._compiled.jnt_type[jid] == mjcf.JNT_FREE
            qa, da = int(self._compiled.jnt_qposadr[jid]), int(self._compiled.jnt_dofadr[jid])
            qpos_indices.extend(range(qa, qa + (7 if free else 1)))
            qvel_indices.extend(range(da, da + (6 if free else 1)))
        return qpos_indices, qvel_indices

    def get_observation_space_mujoco(self, agent: str) -> dict:
        find = xmldict.find_in_nested_dict
        agent_dict = find(self.xml_dict, name=agent, filter_key="@name")
        agent_sites = find(agent_dict, parent="site")
        sensor_dict = find(self.xml_dict, parent="sensor")
        world_dict = find(self.xml_dict, parent="worldbody")
        joint_dict = find(world_dict, parent="joint")
        if self.agent_cameras:
            self.rgb_sensors[agent] = [cam["@name"] for cam in find(agent_dict, parent="camera")]
        indices = self._create_sensor_index_dict(sensor_dict)
        qpos_indices, qvel_indices = self.retrieve_mujoco_all_indices(joint_dict)
        agent_indices, agent_sensors = process_sensors(indices, agent_sites)
        self.agents_observation_index[agent] = {"sensors": agent_indices, "qpos": qpos_indices, "qvel": qvel_indices}
        space = create_sensor_observation_space(agent_sensors)
        for _ in range(len(qpos_indices) + len(qvel_indices)):
            space["low"].append(-np.inf)
            space["high"].append(np.inf)
        return space

    def get_action_space_mujoco(self, agent: str) -> dict:
        find = xmldict.find_in_nested_dict
        space = {"low": [], "high": []}
        agent_dict = find(self.xml_dict, name=agent, filter_key="@name")
        if self.free_joint:
            try:
                free_joint = agent_dict[0]["joint"]
            except (ValueError, KeyError, IndexError):
                raise Exception(f"The agent {agent} has to have a free joint")
            if free_joint["@type"] != "free":
                raise Exception(f"The joint of agent {agent} has to be of type free")
            dof = int(self._compiled.jnt_dofadr[self._compiled.name2id("joint", free_joint["@name"])])
            space["low"] = [-1, -1, -1]
            space["high"] = [1, 1, 1]
            self.agents_action_index[agent] = [dof, dof + 1, dof + 5]
            return space
        actuator_dict = find(self.xml_dict, parent="actuator")
        indices = []
        for joint in find(agent_dict, parent="joint"):
            for motor in find(self.xml_dict, parent="motor", filter_key="@joint", name=joint["@name"]):
                indices.append(actuator_dict[0]["motor"].index(motor))
                low, high = motor["@ctrlrange"].split(" ")
                space["low"].append(float(low))
                space["high"].append(float(high))
        self.agents_action_index[agent] = indices
        return space

    def _upload_tables(self, agents):
        """Hand the gather / scatter tables to the device (mjrl_set_gather_tables / mjrl_set_scatter_tables)."""
        obs = [self.agents_observation_index[a] for a in agents]
        self._handle.set_gather_tables([o["sensors"] for o in obs], [o["qpos"] for o in obs], [o["qvel"] for o in obs])
        self._handle.set_scatter_tables([self.agents_action_index[a] for a in agents], 1 if self.free_joint else 0)
        self._table_agents = list(agents)
        self._obs_len = {a: len(o["sensors"]) + len(o["qpos"]) + len(o["qvel"]) for a, o in zip(agents, obs)}

    # ------------------------------------------------------------------ stepping
    def _actions_to_array(self, actions: dict) -> np.ndarray:
        agents = self._table_agents
        n_phys = max((len(self.agents_action_index[a]) for a in agents), default=0)
        arr = np.zeros((self.n_env, len(agents), max(n_phys, 1)), np.float64)
        for k, agent in enumerate(agents):
            if agent not in actions:
                raise Exception(f"No action for agent {agent}")
            act = np.asarray(actions[agent], dtype=np.float64)
            act = act.reshape(self.n_env, -1) if act.ndim <= 1 or self.n_env > 1 else act
            need = len(self.agents_action_index[agent])
            if act.shape[-1] < need:
                raise Exception(f"The number of actions for agent {agent} is not correct.")
            arr[:, k, :need] = act[:, :need]
        return arr

    def apply_action(self, actions: dict, skip_frames: int = 1):
        """Scatter the actions and advance every copy ``skip_frames`` physics steps (mujoco_parent.py:316-336).
        The same launch also gathers the post-step observations, which ``get_observations`` then serves."""
        arr = self._actions_to_array(actions)
        n_agent, obs_dim = len(self._table_agents), self._handle.size("obs_dim")
        self._skip_frames_hint = skip_frames
        if self._handle.io_agent < 0 and not self._handle.obs_f32:
            # the handle's pinned host buffers: the kernel reads the actions and writes the results over PCIe itself -- no
            # copy in, no copies out, one synchronisation (at numEnvs = 1 the four staged copies of mjrl_step_host were a
            # third of a step's wall time); what is handed on are copies, the buffers are the next step's too
            act_dim = int(arr.shape[-1])
            pin = self._pinned          # (ONE cache of the handle's buffers for every user: a re-sized buffer frees the old one)
            if pin is None or pin[0] is not self._handle or pin[1].shape != arr.shape:
                pin = self._pinned = (self._handle,) + self._handle.host_buffers(act_dim)
            np.copyto(pin[1], arr)
            self._handle.step_pinned(act_dim, skip_frames)
            obs, trunc = pin[2].copy(), pin[5].copy()
        else:
            obs = np.zeros((self.n_env, n_agent, obs_dim), np.float64)
            trunc = np.zeros((self.n_env, n_agent), np.uint8)
            self._handle.step_host(arr, skip_frames, obs=obs, trunc=trunc)
        self.frame += skip_frames
        self._obs_cache = obs
        self._trunc_cache = trunc
        return obs

    def reset(self):
        """mj_resetData + mj_forward for every copy (mujoco_parent.py:341-358)."""
        if isinstance(self.xml_paths, list) and self._variants is None:
            chosen = random.choice(self.xml_paths)
            if chosen != self.xml_path:
                self.xml_path = chosen
                self._init_environment()
                self._after_init_environment()
        self.cap_overflows(report=True)
        self._handle.reset()
        self._episode = None
        if self._variants is not None:
            # every copy drew its own level variant; `xml_path` names copy 0's (with one copy: the reference's attribute)
            self.xml_path = self._variants["paths"][int(self.variant_ids()[0])]
        self._obs_cache = None
        return self.get_sensor_data()

    def variant_ids(self) -> np.ndarray:
        """Index (into the de-duplicated ``xmlPath`` list) of the level variant every copy currently runs."""
        if self._variants is None:
            return np.zeros(self.n_env, np.int32)
        return self._handle.get_field("variant")

    def cap_overflows(self, report: bool = False) -> tuple:
        """(frames cut at nconmax, frames cut at njmax) over all copies since the last call -- MuJoCo's
        mjWARN_CONTACTFULL / mjWARN_CNSTRFULL.  ``reset`` calls this once per episode and warns like MuJoCo does."""
        counts = self._handle.cap_overflows(clear=True)
        if report and any(counts):
            warnings.warn(f"{counts[0]} physics frames ran into nconmax={self._compiled.nconmax} and {counts[1]} into "
                          f"njmax={self._compiled.njmax} during the last episode: contacts were dropped; raise the "
                          f"'nconmax' / 'njmax' config keys", RuntimeWarning, stacklevel=2)
        return counts

    def mujoco_step(self):
        self._handle.step_host(None, 1)
        self._obs_cache = None

    def _squeeze(self, arr):
        return arr[0] if self.n_env == 1 else arr

    def get_sensor_data(self, agent: str = None):
        data = self._handle.get_field("sensordata")
        if agent is None:
            return self._squeeze(data)
        picked = data[:, self.agents_observation_index[agent]["sensors"]]
        return list(picked[0]) if self.n_env == 1 else picked

    def get_observations(self, agent: str) -> np.ndarray:
        """sensordata[idx] | qpos[idx] | qvel[idx] of one agent (mujoco_parent.py:380-392)."""
        k = self._table_agents.index(agent)
        if self._obs_cache is None:
            # after a reset / external state change: one forward-only gather launch
            index = self.agents_observation_index
            sens, qpos, qvel = (self._handle.get_field(n) for n in ("sensordata", "qpos", "qvel"))
            self._obs_cache = np.zeros((self.n_env, len(self._table_agents), self._handle.size("obs_dim")))
            for j, a in enumerate(self._table_agents):
                row = np.concatenate([sens[:, index[a]["sensors"]], qpos[:, index[a]["qpos"]], qvel[:, index[a]["qvel"]]], axis=1)
                self._obs_cache[:, j, :row.shape[1]] = row
        return self._squeeze(self._obs_cache[:, k, :self._obs_len[agent]].copy())

    # ------------------------------------------------------------------ plugin query helpers
    def _folded_xipos(self, name):
        """Inertial position of a static body the model compiler folded into the world (a constant of the model)."""
        return self._compiled.folded_bodies[name]["xipos"]

    def _folded_mass(self, name):
        return self._compiled.folded_bodies[name]["mass"]

    def get_data(self, name: str) -> dict:
        """Body (xipos, mass, euler zyx deg) or geom record (mujoco_parent.py:394-426)."""
        names = self._compiled.names
        if name in getattr(self._compiled, "folded_bodies", {}):
            body = self.data.body(name)
            xmat = np.asarray(body.xmat).reshape(-1, 9)
            euler = np.stack([mat2euler_scipy(x) for x in xmat])
            return {"position": body.xipos, "mass": np.array([self._folded_mass(name)]), "orientation": self._squeeze(euler),
                    "id": body.id, "name": name, "type": "body"}
        if name in names["body"]:
            body = self.data.body(name)
            xmat = np.asarray(body.xmat).reshape(-1, 9)
            euler = np.stack([mat2euler_scipy(x) for x in xmat])
            return {"position": body.xipos, "mass": self.model.body(name).mass, "orientation": self._squeeze(euler),
                    "id": body.id, "name": name, "type": "body"}
        if name in names["geom"]:
            geom = self.data.geom(name)
            xmat = np.asarray(geom.xmat).reshape(-1, 9)
            euler = np.stack([mat2euler_scipy(x) for x in xmat])
            mg = self.model.geom(name)
            return {"position": geom.xpos, "orientation": self._squeeze(euler), "id": geom.id, "name": name,
                    "type": "geom", "color": mg.rgba, "shape": mg.type}
        raise KeyError(f"Invalid name '{name}'")

    def distance(self, object_1, object_2):
        """Euclidean distance between two objects given by name (body xipos, else geom xpos) or by coordinates
        (mujoco_parent.py:428-449)."""
        def coordinates(obj):
            if isinstance(obj, str):
                if obj in self._compiled.names["body"] or obj in getattr(self._compiled, "folded_bodies", {}):
                    return np.asarray(self.data.body(obj).xipos)
                return np.asarray(self.data.geom(obj).xpos)
            return np.asarray(obj, dtype=np.float64)
        a, b = coordinates(object_1), coordinates(object_2)
        if a.ndim == 1 and b.ndim == 1:
            return math.dist(a, b)
        return np.linalg.norm(a - b, axis=-1)

    def collision(self, geom_1, geom_2):
        """True when the two geoms share a contact (mujoco_parent.py:451-478)."""
        ids = []
        for g in (geom_1, geom_2):
            if isinstance(g, str):
                if g not in self._compiled.names["geom"]:
                    raise Exception(f"Collision object {g} not found in data")
                g = self._compiled.names["geom"].index(g)
            ids.append(int(g))
        pairs = self._handle.query("contact_geom").astype(int)
        hit = ((pairs[..., 0] == ids[0]) & (pairs[..., 1] == ids[1])) | ((pairs[..., 0] == ids[1]) & (pairs[..., 1] == ids[0]))
        hit = hit.any(axis=1)
        return bool(hit[0]) if self.n_env == 1 else hit

    def get_camera_data(self, cam_object: str):
        """Images of all cameras of an agent, ``(ncam_agent, W, H, 3)`` uint8, or of one camera by name
        (mujoco_parent.py:540-555); with several env copies a leading ``[numEnvs]`` axis is added.  Ray cast on
        the device (mjrl_render_*), not OpenGL."""
        width, height = self.sensor_resolution
        names = self._compiled.names["camera"]
        asked = self.rgb_sensors.get(cam_object, [cam_object] if cam_object in names else [])
        for cam in asked:
            mode = self._compiled.camera_mode[names.index(cam)]
            if mode != "fixed":
                raise Exception(f'camera {cam} has mode="{mode}"; the ray caster draws cameras fixed in their body only')
        images = self._handle.render(width, height)            # [n_env, ncam, H, W, 3], rows bottom-up
        if not self._scene_cache and names:
            # (a camera asked for by name on a handle made without agentCameras: this image was drawn from the current
            # qpos; from the next step on the step keeps its frames and the images follow the reference's rule)
            self._handle.set_scene_cache(True)
            self._scene_cache = True
        images = images.reshape(self.n_env, len(names), width, height, 3)   # the reference's (W, H, 3) view of it
        if cam_object in self.rgb_sensors:
            picked = images[:, [names.index(c) for c in self.rgb_sensors[cam_object]]]
        elif cam_object in names:
            picked = images[:, names.index(cam_object)]
        else:
            raise Exception(f"{cam_object} is neither an agent with cameras (agentCameras must be set) nor a camera")
        return picked[0] if self.n_env == 1 else picked

    def start_render(self):
        raise Exception("interactive rendering is out of scope of the batched stepper")

    def end_render(self):
        self.render = False
