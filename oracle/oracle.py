"""ctypes front end of the CPU oracle (oracle/ora_step.c).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package never does.  Physics parity is unpinned (see ora_math.h).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmjrl_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("ora_step.c", "ora_math.h", "ora_collide.h", "ora_layout.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.ora_model_create.restype = ctypes.c_void_p
        L.ora_model_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.ora_model_destroy.argtypes = [ctypes.c_void_p]
        L.ora_data_create.restype = ctypes.c_void_p
        L.ora_data_create.argtypes = [ctypes.c_void_p]
        L.ora_data_destroy.argtypes = [ctypes.c_void_p]
        for fn in ("ora_forward", "ora_step", "ora_reset"):
            getattr(L, fn).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ora_field.restype = ctypes.POINTER(ctypes.c_double)
        L.ora_field.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        for fn in ("ora_ncon", "ora_nefc", "ora_niter", "ora_warnings"):
            getattr(L, fn).argtypes = [ctypes.c_void_p]
            getattr(L, fn).restype = ctypes.c_int
        L.ora_time.argtypes = [ctypes.c_void_p]
        L.ora_time.restype = ctypes.c_double
        L.ora_contact_get.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        L.ora_render.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.ora_model_size.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.ora_model_size.restype = ctypes.c_int
        _lib = L
    return _lib


_FIELD_SHAPES = {
    "qpos": "nq", "qvel": "nv", "ctrl": "nu", "qacc_warmstart": "nv",
    "xpos": ("nbody", 3), "xquat": ("nbody", 4), "xmat": ("nbody", 9), "xipos": ("nbody", 3), "ximat": ("nbody", 9),
    "xanchor": ("njnt", 3), "xaxis": ("njnt", 3), "geom_xpos": ("ngeom", 3), "geom_xmat": ("ngeom", 9),
    "site_xpos": ("nsite", 3), "site_xmat": ("nsite", 9), "cam_xpos": ("ncam", 3), "cam_xmat": ("ncam", 9),
    "subtree_com": ("nbody", 3), "cinert": ("nbody", 10), "crb": ("nbody", 10), "cdof": ("nv", 6),
    "cdof_dot": ("nv", 6), "cvel": ("nbody", 6), "cacc": ("nbody", 6), "cfrc": ("nbody", 6),
    "qM": "nM", "qLD": "nM", "qLDiagInv": "nv", "qMdense": ("nv", "nv"),
    "qfrc_bias": "nv", "qfrc_passive": "nv", "qfrc_actuator": "nv", "qfrc_smooth": "nv", "qacc_smooth": "nv",
    "qfrc_constraint": "nv", "qacc": "nv", "sensordata": "nsensordata",
    "efc_J": ("njmax", "nv"), "efc_pos": "njmax", "efc_margin": "njmax", "efc_diagApprox": "njmax",
    "efc_R": "njmax", "efc_D": "njmax", "efc_vel": "njmax", "efc_aref": "njmax", "efc_b": "njmax",
    "efc_force": "njmax", "efc_KBIP": ("njmax", 4),
}


class OracleEnv:
    """One env copy stepped by the CPU oracle.  Arrays are live numpy views of the C buffers."""

    def __init__(self, blob: bytes):
        L = lib()
        self._blob = blob
        self._m = L.ora_model_create(blob, len(blob))
        if not self._m:
            raise RuntimeError("oracle rejected the model blob (magic/version/size mismatch)")
        self._d = L.ora_data_create(self._m)
        self._views = {}
        L.ora_reset(self._m, self._d)

    def size(self, name: str) -> int:
        return lib().ora_model_size(self._m, name.encode())

    def __getattr__(self, name):
        if name in _FIELD_SHAPES:
            if name not in self._views:
                shape = _FIELD_SHAPES[name]
                shape = shape if isinstance(shape, tuple) else (shape,)
                shape = tuple(self.size(s) if isinstance(s, str) else s for s in shape)
                n = int(np.prod(shape)) if shape else 1
                ptr = lib().ora_field(self._d, name.encode())
                self._views[name] = (np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].reshape(shape)
                                     if n else np.zeros(shape))
            return self._views[name]
        raise AttributeError(name)

    def reset(self):
        lib().ora_reset(self._m, self._d)

    def forward(self):
        lib().ora_forward(self._m, self._d)

    def step(self, n: int = 1):
        L = lib()
        for _ in range(n):
            L.ora_step(self._m, self._d)

    @property
    def ncon(self):
        return lib().ora_ncon(self._d)

    @property
    def nefc(self):
        return lib().ora_nefc(self._d)

    @property
    def niter(self):
        return lib().ora_niter(self._d)

    @property
    def warnings(self):
        return lib().ora_warnings(self._d)

    @property
    def time(self):
        return lib().ora_time(self._d)

    def render(self, cam: int, width: int = 64, height: int = 64):
        """uint8 image (width, height, 3) of fixed camera ``cam`` (rows bottom-up), drawn from the frames the last
        forward pass left in the data (step, forward or reset) -- what mjv_updateScene would read."""
        out = np.zeros((height, width, 3), np.uint8)
        lib().ora_render(self._m, self._d, cam, width, height, out.ctypes.data_as(ctypes.c_void_p))
        return out.reshape(width, height, 3)

    def render_hits(self, cam: int, width: int = 64, height: int = 64):
        """Per pixel of ``render``'s image: the geom its ray hits (-1: none) and the distance (test harness)."""
        hit = np.zeros((height, width), np.int32)
        dist = np.zeros((height, width), np.float64)
        lib().ora_render_hits.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
        lib().ora_render_hits(self._m, self._d, cam, width, height, hit.ctypes.data_as(ctypes.c_void_p),
                              dist.ctypes.data_as(ctypes.c_void_p))
        return hit, dist

    def contacts(self):
        out = []
        buf = (ctypes.c_double * 18)()
        for i in range(self.ncon):
            lib().ora_contact_get(self._d, i, buf)
            v = np.array(buf[:])
            out.append(dict(dist=v[0], pos=v[1:4].copy(), frame=v[4:13].reshape(3, 3).copy(), includemargin=v[13],
                            geom1=int(v[14]), geom2=int(v[15]), efc_address=int(v[16]), normal_force=v[17]))
        return out

    def close(self):
        if getattr(self, "_d", None):
            lib().ora_data_destroy(self._d)
            lib().ora_model_destroy(self._m)
            self._d = self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
