"""ctypes binding of libmjrl_hip.so (include/mjrl.h).

This is the only native library the package loads.  There is no CPU fallback: if the HIP library is
missing or fails to load, importing this module's ``load()`` raises.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MJRL_LIB") or os.path.join(_HERE, "csrc", "libmjrl_hip.so")     # (MJRL_LIB: experiments, A/B of two builds)

# every symbol include/mjrl.h declares
SYMBOLS = [
    "mjrl_version", "mjrl_last_error", "mjrl_create", "mjrl_destroy", "mjrl_set_stream", "mjrl_sync",
    "mjrl_set_gather_tables", "mjrl_set_scatter_tables", "mjrl_set_max_steps", "mjrl_size", "mjrl_reset",
    "mjrl_step_device", "mjrl_step_host", "mjrl_get_field", "mjrl_set_field", "mjrl_query", "mjrl_step_debug",
    "mjrl_lds_offset", "mjrl_step_profile", "mjrl_set_program", "mjrl_set_query_cache", "mjrl_set_scene_cache",
    "mjrl_render_device", "mjrl_render_host", "mjrl_load_kernel", "mjrl_cap_overflows", "mjrl_step_timeline",
    "mjrl_step_truncated", "mjrl_host_buffers", "mjrl_step_pinned", "mjrl_set_autoreset",
    "mjrl_reset_device", "mjrl_set_step_reset_mask", "mjrl_set_tag_tables", "mjrl_set_env_base", "mjrl_set_variants",
    "mjrl_encoder_load", "mjrl_encode_device", "mjrl_encode_host", "mjrl_set_camera_obs", "mjrl_set_io_layout",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
            "(or __graft_entry__.build()).  The stepper has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise RuntimeError(f"libmjrl_hip.so lacks symbols declared in include/mjrl.h: {missing}")
    vp, ip, sz, ci = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_size_t, ctypes.c_int
    L.mjrl_version.restype = ctypes.c_char_p
    L.mjrl_last_error.restype = ctypes.c_char_p
    L.mjrl_last_error.argtypes = [vp]
    L.mjrl_create.argtypes = [ctypes.c_char_p, sz, ci, ci, ctypes.c_uint, ctypes.POINTER(vp)]
    L.mjrl_destroy.argtypes = [vp]
    L.mjrl_destroy.restype = None
    L.mjrl_set_stream.argtypes = [vp, vp]
    L.mjrl_sync.argtypes = [vp]
    L.mjrl_set_gather_tables.argtypes = [vp, ci, ip, ip, ip, ip, ip, ip]
    L.mjrl_set_scatter_tables.argtypes = [vp, ci, ci, ip, ip]
    L.mjrl_set_max_steps.argtypes = [vp, ci]
    L.mjrl_size.argtypes = [vp, ctypes.c_char_p]
    L.mjrl_reset.argtypes = [vp, vp, vp]
    L.mjrl_reset_device.argtypes = [vp, vp, vp]
    L.mjrl_set_step_reset_mask.argtypes = [vp, vp]
    L.mjrl_set_tag_tables.argtypes = [vp, ci, ip, ip]
    L.mjrl_set_env_base.argtypes = [vp, ci]
    L.mjrl_set_variants.argtypes = [vp, ci, vp, ctypes.c_ulonglong]
    L.mjrl_encoder_load.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp]
    L.mjrl_encode_device.argtypes = [vp, vp, ci, vp]
    L.mjrl_encode_host.argtypes = [vp, vp, ci, vp]
    L.mjrl_set_camera_obs.argtypes = [vp, ci, ip]
    L.mjrl_step_device.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    L.mjrl_step_host.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    L.mjrl_get_field.argtypes = [vp, ctypes.c_char_p, vp, sz]
    L.mjrl_set_field.argtypes = [vp, ctypes.c_char_p, vp, sz]
    L.mjrl_query.argtypes = [vp, ctypes.c_char_p, vp, sz]
    L.mjrl_step_debug.argtypes = [vp, vp, ci, ci, ci, vp, sz]
    L.mjrl_lds_offset.argtypes = [vp, ctypes.c_char_p]
    L.mjrl_step_profile.argtypes = [vp, vp, ci, ci, vp, ci]
    L.mjrl_set_program.argtypes = [vp, ci, ip, vp, ci, ci, ip]
    L.mjrl_set_query_cache.argtypes = [vp, ci]
    L.mjrl_set_scene_cache.argtypes = [vp, ci]
    L.mjrl_render_device.argtypes = [vp, ci, ci, vp]
    L.mjrl_render_host.argtypes = [vp, ci, ci, vp]
    L.mjrl_load_kernel.argtypes = [vp, ctypes.c_char_p]
    L.mjrl_cap_overflows.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    L.mjrl_step_timeline.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t]
    L.mjrl_step_truncated.argtypes = [vp, ctypes.c_int]
    L.mjrl_host_buffers.argtypes = [vp, ctypes.c_int] + [ctypes.POINTER(ctypes.c_void_p)] * 5
    L.mjrl_step_pinned.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.mjrl_set_autoreset.argtypes = [vp, ctypes.c_int]
    L.mjrl_set_io_layout.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    _lib = L
    return L


def _i32(seq):
    arr = np.ascontiguousarray(np.asarray(seq, dtype=np.int32).reshape(-1))
    if arr.size == 0:
        arr = np.zeros(1, np.int32)
    return arr, arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _host_ptr(arr):
    return None if arr is None else ctypes.c_void_p(arr.ctypes.data)


class Handle:
    """Owns one ``mjrl_env`` (one GPU, n_env copies of one model)."""

    def __init__(self, blob: bytes, n_env: int, device_id: int = 0, specialize: bool | None = None,
                 few: bool | None = None):
        """``few``: None = the library's rule (a batch of at most one wave per SIMD gets the build for such batches,
        which assumes that this handle has the device to itself); False = other handles step beside this one, use the
        full-batch build; True = the few-copies build whatever the batch size (mjrl_create flags bits 1 / 2)."""
        self._lib = load()
        self._h = ctypes.c_void_p()
        # (MJRL_CREATE_FLAGS: experiments only -- bit 0 switches the longest-first dispatch off)
        flags = int(os.environ.get("MJRL_CREATE_FLAGS", "0"))
        if few is not None:
            flags |= 4 if few else 2
        rc = self._lib.mjrl_create(blob, len(blob), int(n_env), int(device_id), flags, ctypes.byref(self._h))
        if rc:
            raise Exception(f"mjrl_create failed ({rc}): {self._lib.mjrl_last_error(None).decode()}")
        self.n_env = int(n_env)
        self.io_agent, self.obs_f32 = -1, False
        self.kernel = "generic"
        if specialize is None:
            specialize = os.environ.get("MJRL_SPECIALIZE", "1") != "0"
        if specialize:
            from . import kernel_cache
            path = kernel_cache.code_object(blob, few=bool(self.size("few")))
            if path is not None:
                self.load_kernel(path)

    def load_kernel(self, path: str | None):
        """Attach a model-specialised step kernel (``kernel_cache.code_object``); ``None`` returns to the generic one."""
        self._check(self._lib.mjrl_load_kernel(self._h, path.encode() if path else None))
        self.kernel = "specialised" if path else "generic"

    def cap_overflows(self, clear: bool = False) -> tuple[int, int]:
        """(frames cut at nconmax, frames cut at njmax), summed over the env copies since creation / the last clear --
        MuJoCo's mjWARN_CONTACTFULL / mjWARN_CNSTRFULL counts."""
        out = (ctypes.c_ulonglong * 2)()
        self._check(self._lib.mjrl_cap_overflows(self._h, out, 1 if clear else 0))
        return int(out[0]), int(out[1])

    def _check(self, rc):
        if rc:
            raise Exception(f"libmjrl_hip error {rc}: {self._lib.mjrl_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mjrl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self, name: str) -> int:
        return self._lib.mjrl_size(self._h, name.encode())

    def lds_offset(self, region: str) -> int:
        return self._lib.mjrl_lds_offset(self._h, region.encode())

    def set_stream(self, hip_stream: int | None):
        self._check(self._lib.mjrl_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)))

    def sync(self):
        self._check(self._lib.mjrl_sync(self._h))

    def render(self, width: int, height: int, d_rgb: int | None = None):
        """uint8 ``[n_env, ncam, height, width, 3]``; with ``d_rgb`` (device address) nothing is copied back."""
        if d_rgb:
            self._check(self._lib.mjrl_render_device(self._h, int(width), int(height), ctypes.c_void_p(d_rgb)))
            return None
        out = np.zeros((self.n_env, self.size("ncam"), height, width, 3), np.uint8)
        self._check(self._lib.mjrl_render_host(self._h, int(width), int(height), _host_ptr(out)))
        return out

    def set_query_cache(self, enabled: bool):
        self._check(self._lib.mjrl_set_query_cache(self._h, int(bool(enabled))))

    def set_scene_cache(self, enabled: bool):
        """Render what the last forward pass left (after a step: frames one integration older than qpos, what
        mjv_updateScene reads out of MjData) instead of fresh kinematics of the current qpos."""
        self._check(self._lib.mjrl_set_scene_cache(self._h, int(bool(enabled))))

    def set_max_steps(self, n: int):
        self._check(self._lib.mjrl_set_max_steps(self._h, int(n)))

    def set_gather_tables(self, sensor_idx, qpos_idx, qvel_idx):
        """Each argument: list (one entry per agent) of index lists."""
        n_agent = len(sensor_idx)
        keep = []
        args = []
        for lists in (sensor_idx, qpos_idx, qvel_idx):
            counts, cp = _i32([len(x) for x in lists])
            flat, fp = _i32([i for x in lists for i in x])
            keep += [counts, flat]
            args += [cp, fp]
        self._check(self._lib.mjrl_set_gather_tables(self._h, n_agent, *args))

    def set_scatter_tables(self, idx_lists, mode: int):
        counts, cp = _i32([len(x) for x in idx_lists])
        flat, fp = _i32([i for x in idx_lists for i in x])
        self._check(self._lib.mjrl_set_scatter_tables(self._h, len(idx_lists), int(mode), cp, fp))

    def set_program(self, prog_i, prog_f, n_slot: int, n_extra_obs: int, agent_body):
        pi = np.ascontiguousarray(np.asarray(prog_i, dtype=np.int32).reshape(-1, 8))
        pf = np.ascontiguousarray(np.asarray(prog_f, dtype=np.float64).reshape(-1, 4))
        if pi.shape[0] != pf.shape[0]:
            raise Exception("program: integer and float parts differ in length")
        n_op = pi.shape[0]
        if n_op == 0:
            pi, pf = np.zeros((1, 8), np.int32), np.zeros((1, 4))
        body, bp = _i32(agent_body)
        self._check(self._lib.mjrl_set_program(self._h, n_op, pi.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                               _host_ptr(pf), int(n_slot), int(n_extra_obs), bp))

    def reset(self, mask=None, d_obs: int | None = None):
        m = None
        if mask is not None:
            m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
            if m.size != self.n_env:
                raise Exception(f"reset mask has {m.size} entries for {self.n_env} env copies")
        self._check(self._lib.mjrl_reset(self._h, _host_ptr(m), ctypes.c_void_p(d_obs or 0)))

    def set_tag_tables(self, tags):
        """``tags``: list (one entry per tag, in tag-index order) of lists of (kind, id) pairs, kind 0 body / 1 geom."""
        counts, cp = _i32([len(t) for t in tags])
        flat, fp = _i32([(k << 16) | i for t in tags for k, i in t])
        self._check(self._lib.mjrl_set_tag_tables(self._h, len(tags), cp, fp))

    def set_env_base(self, first_env_id: int):
        self._check(self._lib.mjrl_set_env_base(self._h, int(first_env_id)))

    def set_variants(self, rgba, seed: int = 0):
        """``rgba``: ``[n_variant, ngeom, 4]`` colour tables (None / empty: off)."""
        if rgba is None or len(rgba) == 0:
            self._check(self._lib.mjrl_set_variants(self._h, 0, None, 0))
            return
        arr = np.ascontiguousarray(np.asarray(rgba, dtype=np.float64))
        if arr.ndim != 3 or arr.shape[1:] != (self.size("ngeom"), 4):
            raise Exception(f"variant colours have shape {arr.shape}, expected (n_variant, {self.size('ngeom')}, 4)")
        self._check(self._lib.mjrl_set_variants(self._h, arr.shape[0], _host_ptr(arr), int(seed)))

    def encoder_load(self, weights: dict, relu: bool = True):
        """``weights``: w1 [3,3,3,32], b1 [32], w2 [3,3,32,64], b2 [64], wd [16384, latent], bd [latent] (Keras layout)."""
        shapes = {"w1": (3, 3, 3, 32), "b1": (32,), "w2": (3, 3, 32, 64), "b2": (64,)}
        arrs = {k: np.ascontiguousarray(np.asarray(weights[k], dtype=np.float32)) for k in ("w1", "b1", "w2", "b2", "wd", "bd")}
        for k, shape in shapes.items():
            if arrs[k].shape != shape:
                raise Exception(f"encoder weight {k} has shape {arrs[k].shape}, expected {shape}")
        latent = arrs["bd"].shape[0]
        if arrs["wd"].shape != (16384, latent):
            raise Exception(f"encoder weight wd has shape {arrs['wd'].shape}, expected (16384, {latent})")
        self._check(self._lib.mjrl_encoder_load(self._h, latent, int(bool(relu)), *[_host_ptr(arrs[k]) for k in ("w1", "b1", "w2", "b2", "wd", "bd")]))
        self.latent_dim = latent

    def encode(self, images=None, d_rgb: int | None = None, n_img: int | None = None, d_latent: int | None = None):
        """Latents of uint8 images ``[n, 64, 64, 3]`` (host array -> host float32 array), or of a device buffer."""
        if d_rgb:
            self._check(self._lib.mjrl_encode_device(self._h, ctypes.c_void_p(d_rgb), int(n_img), ctypes.c_void_p(d_latent)))
            return None
        img = np.ascontiguousarray(np.asarray(images, dtype=np.uint8)).reshape(-1, 64, 64, 3)
        out = np.zeros((img.shape[0], self.size("latent_dim")), np.float32)
        self._check(self._lib.mjrl_encode_host(self._h, _host_ptr(img), img.shape[0], _host_ptr(out)))
        return out

    def set_camera_obs(self, agent_cam):
        """``agent_cam``: camera id per agent (-1: none); None / empty turns the camera observation off."""
        if not agent_cam:
            self._check(self._lib.mjrl_set_camera_obs(self._h, 0, None))
            return
        arr, ptr = _i32(agent_cam)
        self._check(self._lib.mjrl_set_camera_obs(self._h, len(agent_cam), ptr))

    def reset_device(self, d_mask: int | None, d_obs: int | None = None):
        """Reset the copies flagged in the device byte mask at address ``d_mask`` (None: all); asynchronous."""
        self._check(self._lib.mjrl_reset_device(self._h, ctypes.c_void_p(d_mask or 0), ctypes.c_void_p(d_obs or 0)))

    def set_step_reset_mask(self, d_mask: int | None):
        """Every later step launch first resets the copies flagged in the device byte mask (None: off)."""
        self._check(self._lib.mjrl_set_step_reset_mask(self._h, ctypes.c_void_p(d_mask or 0)))

    def set_autoreset(self, mode: int):
        """0 off; 1: a copy whose episode ended is reset by the next step without being stepped (Gymnasium next-step
        autoreset); 2: it is reset and stepped in that launch (the reference's ``env.reset(); env.step(a)``)."""
        self._check(self._lib.mjrl_set_autoreset(self._h, int(mode)))

    def set_io_layout(self, agent: int = -1, obs_f32: bool = False):
        """One agent's rows only (``actions [n_env, act_dim]``, ``obs [n_env, obs_dim]``) and / or float32 observations
        for the device and pinned step entries (mjrl_set_io_layout); ``(-1, False)`` is the default layout."""
        self._check(self._lib.mjrl_set_io_layout(self._h, int(agent), int(bool(obs_f32))))
        self.io_agent, self.obs_f32 = int(agent), bool(obs_f32)

    def step_device(self, d_actions, act_dim, skip_frames, d_obs=None, d_reward=None, d_term=None, d_trunc=None):
        """All pointers are integer device addresses (e.g. ``tensor.data_ptr()``) or None."""
        c = lambda p: ctypes.c_void_p(p or 0)
        self._check(self._lib.mjrl_step_device(self._h, c(d_actions), int(act_dim), int(skip_frames), c(d_obs),
                                               c(d_reward), c(d_term), c(d_trunc)))

    def step_host(self, actions, skip_frames, obs=None, reward=None, term=None, trunc=None):
        act_dim = 0 if actions is None else int(actions.shape[-1])
        self._check(self._lib.mjrl_step_host(self._h, _host_ptr(actions), act_dim, int(skip_frames), _host_ptr(obs),
                                             _host_ptr(reward), _host_ptr(term), _host_ptr(trunc)))

    def host_buffers(self, act_dim: int):
        """Pinned host arrays ``(actions, obs, reward, term, trunc)`` that ``step_pinned`` reads and writes in place
        (numpy views of memory owned by the handle; every step overwrites them)."""
        ptrs = [ctypes.c_void_p() for _ in range(5)]
        self._check(self._lib.mjrl_host_buffers(self._h, int(act_dim), *[ctypes.byref(p) for p in ptrs]))
        n_agent, obs_dim = max(self.size("n_agent"), 1), max(self.size("obs_dim"), 1)
        def view(ptr, ctype, shape):
            n = int(np.prod(shape))
            return np.ctypeslib.as_array((ctype * n).from_address(ptr.value)).reshape(shape)
        # (with a one-agent / float32 layout the kernel uses the head of the same buffers: the views have the layout's shape)
        act_shape = (self.n_env, n_agent, max(int(act_dim), 1)) if self.io_agent < 0 else (self.n_env, max(int(act_dim), 1))
        obs_shape = (self.n_env, n_agent, obs_dim) if self.io_agent < 0 else (self.n_env, obs_dim)
        return (view(ptrs[0], ctypes.c_double, act_shape),
                view(ptrs[1], ctypes.c_float if self.obs_f32 else ctypes.c_double, obs_shape),
                view(ptrs[2], ctypes.c_double, (self.n_env, n_agent)),
                view(ptrs[3], ctypes.c_uint8, (self.n_env, n_agent)),
                view(ptrs[4], ctypes.c_uint8, (self.n_env, n_agent)))

    def step_pinned(self, act_dim: int, skip_frames: int):
        self._check(self._lib.mjrl_step_pinned(self._h, int(act_dim), int(skip_frames)))

    def get_field(self, name: str):
        per = {"qpos": "nq", "qvel": "nv", "ctrl": "nu", "qacc_warmstart": "nv", "sensordata": "nsensordata"}
        if name == "timestep":
            out = np.zeros(self.n_env, np.int32)
        elif name == "solver_stats":
            out = np.zeros((self.n_env, 4), np.int32)
        elif name in ("variant", "episode"):
            out = np.zeros(self.n_env, np.int32)
        elif name == "store":
            out = np.zeros((self.n_env, max(self.size("n_agent"), 1), max(self.size("n_slot"), 0)), np.float64)
        else:
            out = np.zeros((self.n_env, self.size(per[name])), np.float64)
        self._check(self._lib.mjrl_get_field(self._h, name.encode(), _host_ptr(out), out.nbytes))
        return out

    def set_field(self, name: str, value):
        dtype = np.int32 if name in ("timestep", "variant", "episode") else np.float64
        arr = np.ascontiguousarray(np.asarray(value, dtype=dtype))
        self._check(self._lib.mjrl_set_field(self._h, name.encode(), _host_ptr(arr), arr.nbytes))

    def query(self, name: str):
        shapes = {"xpos": ("nbody", 3), "xquat": ("nbody", 4), "xipos": ("nbody", 3), "geom_xpos": ("ngeom", 3),
                  "geom_xmat": ("ngeom", 9), "ncon": (1,), "warn": (1,), "contact_geom": ("nconmax", 2)}
        shape = tuple(self.size(s) if isinstance(s, str) else s for s in shapes[name])
        out = np.zeros((self.n_env,) + shape, np.float64)
        self._check(self._lib.mjrl_query(self._h, name.encode(), _host_ptr(out), out.nbytes))
        return out

    STAGES = ["load", "kin", "com", "crb", "factor", "geom", "collide", "vel", "smooth", "rows", "project", "pgs",
              "sensors", "euler", "store", "pgs_warm", "pgs_lists", "pgs_sweeps", "rows_limits", "rows_addr", "pgs_setup", "tail"]

    def step_profile(self, skip_frames=1):
        out = np.zeros(len(self.STAGES), np.uint64)
        self._check(self._lib.mjrl_step_profile(self._h, None, 0, int(skip_frames), _host_ptr(out), out.size))
        return dict(zip(self.STAGES, out.tolist()))

    def step_truncated(self, stage: str):
        """Diagnostic launch that ends every wave right after ``stage`` (a name of STAGES[:15]); writes nothing back."""
        self._check(self._lib.mjrl_step_truncated(self._h, self.STAGES.index(stage) + 1))

    def step_timeline(self, d_actions=None, act_dim=0):
        """One step; ``[n_env, 3]`` uint64 per workgroup in dispatch order: wave start, wave end (100 MHz ticks), copy."""
        out = np.zeros((self.n_env, 3), np.uint64)
        self._check(self._lib.mjrl_step_timeline(self._h, ctypes.c_void_p(d_actions or 0), int(act_dim), 1,
                                                 _host_ptr(out), out.size))
        return out

    def step_debug(self, d_actions, act_dim, skip_frames, stage=0):
        out = np.zeros((self.n_env, self.size("lds_doubles")), np.float64)
        self._check(self._lib.mjrl_step_debug(self._h, ctypes.c_void_p(d_actions or 0), int(act_dim), int(skip_frames),
                                              int(stage), _host_ptr(out), out.nbytes))
        return out
