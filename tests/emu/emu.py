"""ctypes front end of the CPU lane-emulation build of the device step code.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MJRL_EMU_SANITIZED=1 (set by tests/test_emu_sanitized.py for its child process, which preloads libasan): the
# ASan + UBSan build of the same source
_SANITIZED = os.environ.get("MJRL_EMU_SANITIZED") == "1"
_LIB = os.path.join(_HERE, "_build", "libmjrl_emu_san.so" if _SANITIZED else "libmjrl_emu.so")
_lib = None

REGION_SHAPES = {
    "qpos": ("nq",), "qvel": ("nv",), "ctrl": ("nu",), "warm": ("nv",), "xpos": ("nbody", 3), "xquat": ("nbody", 4),
    "xanchor": ("njnt", 3), "xaxis": ("njnt", 3), "cinert": ("nbody", 10), "crb": ("nbody", 10), "cdof": ("nv", 6),
    "cdofdot": ("nv", 6), "cvel": ("nbody", 6), "cacc": ("nbody", 6), "LD": ("nM",), "Dinv": ("nv",),
    "gpos": ("ngeom", 3), "gquat": ("ngeom", 4), "bias": ("nv",), "smooth": ("nv",), "qaccs": ("nv",), "x": ("nv",),
    "qfc": ("nv",), "qacc": ("nv",), "con": ("nconmax", 15), "row": ("njmax", 4), "sens": ("nsensordata",),
}


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-C", _HERE, "-s"] + (["sanitized"] if _SANITIZED else []), check=True)
        L = ctypes.CDLL(_LIB)
        L.emu_lds_total.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.emu_lds_offset.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        P = ctypes.c_void_p
        L.emu_step.argtypes = [ctypes.c_char_p, ctypes.c_size_t, P, P, P, P, P, P, P, P, ctypes.c_int, ctypes.c_int,
                               ctypes.c_int, P, ctypes.c_int, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P,
                               ctypes.c_int, ctypes.c_int, P, P, ctypes.c_int, ctypes.c_int, P, P, P, P, P, P]
        L.emu_set_step_reset.argtypes = [ctypes.c_int, P]
        L.emu_set_step_reset.restype = None
        L.emu_set_reset_sens.argtypes = [P]
        L.emu_set_reset_sens.restype = None
        L.emu_set_autoreset.argtypes = [P, ctypes.c_int, P]
        L.emu_set_autoreset.restype = None
        L.emu_set_tags.argtypes = [P, P, P, ctypes.c_int]
        L.emu_set_tags.restype = None
        L.emu_set_io_layout.argtypes = [ctypes.c_int, ctypes.c_int]
        L.emu_set_io_layout.restype = None
        L.emu_set_few.argtypes = [ctypes.c_int]
        L.emu_set_few.restype = None
        L.emu_set_lpt.argtypes = [P, P, ctypes.c_int]
        L.emu_lpt_lookup.argtypes = [P, P, ctypes.c_int, ctypes.c_int, P, ctypes.c_int, P]
        L.emu_set_lpt.restype = None
        _lib = L
    return _lib


def _p(arr):
    return None if arr is None else arr.ctypes.data_as(ctypes.c_void_p)


class LdsImage:
    """View of one env copy's LDS image dumped by the step code."""

    def __init__(self, image, model, offset_fn):
        self.image, self.model, self._off = image, model, offset_fn
        i0 = self._off("ints")
        self.ints = image[i0:].view(np.int32)

    def region(self, name):
        shape = tuple(getattr(self.model, s) if isinstance(s, str) else s for s in REGION_SHAPES[name])
        n = int(np.prod(shape))
        o = self._off(name)
        return self.image[o:o + n].reshape(shape)

    @property
    def ncon(self): return int(self.ints[0])
    @property
    def nefc(self): return int(self.ints[1])
    @property
    def niter(self): return int(self.ints[3])
    @property
    def warn(self): return int(self.ints[4])

    def J(self):
        """The constraint rows expanded from their stored forms (mjrl_step.h, JW: tree-local or compact chains)."""
        m = self.model
        ldj, o, info_at = self._off("ldj"), self._off("J"), self._off("i_rowinfo")
        rows = self.image[o:o + ldj * m.njmax].reshape(m.njmax, ldj)
        parent = m.dof_parentid
        dense = np.zeros((m.njmax, m.nv))
        for r in range(self.nefc):
            info = int(self.ints[info_at + r])
            tree = (info >> 19) - 2
            if m.rowmap and tree >= 0:          # slot i = the tree's i-th dof
                a, n = int(m.tree_dofadr[tree]), int(m.tree_dofnum[tree])
                dense[r, a:a + n] = rows[r, :n]
                continue
            xp, xq1 = info & 63, (info >> 9) & 127
            seen = set()
            d, t = xp, 0
            while d >= 0:                       # primary chain: slots 0.. from the deepest dof up
                dense[r, d] = rows[r, t]; seen.add(d)
                d, t = int(parent[d]), t + 1
            d, t = xq1 - 1, 0
            while xq1 and d >= 0:               # secondary chain: slots 8.., shared ancestors live on the primary one
                if d not in seen:
                    dense[r, d] = rows[r, 8 + t]
                d, t = int(parent[d]), t + 1
        return dense

    def M_from_factor(self):
        """The sparse inertia matrix (dof_Madr layout) rebuilt from its L'DL factor: M[i][j] = sum over the dofs k at or
        below i of L[k][i] D[k] L[k][j] (the kernel keeps the unfactorised M in an HBM scratch buffer, not in LDS)."""
        m = self.model
        ld, madr, depth, parent = self.region("LD"), m.dof_Madr, m.dof_depth, m.dof_parentid
        out = np.zeros(m.nM)
        for k in range(m.nv):
            chain, d = [], k
            while d >= 0:
                chain.append(d); d = int(parent[d])
            Dk = ld[madr[k]]
            Lk = [1.0] + [ld[madr[k] + t] for t in range(1, int(depth[k]) + 1)]      # L[k][chain[t]]
            for ti, i in enumerate(chain):
                for tj in range(ti, len(chain)):                                    # chain[tj] is an ancestor of i
                    out[madr[i] + (tj - ti)] += Lk[ti] * Dk * Lk[tj]
        return out

    def contact_geoms(self):
        a, b = self._off("i_cong1"), self._off("i_cong2")
        return np.stack([self.ints[a:a + self.ncon], self.ints[b:b + self.ncon]], axis=1)


class EmuEnv:
    @staticmethod
    def _program_args(program):
        if program is None:
            return (None, None, 0, 0, None, None, None, None, None, None)
        pi = np.ascontiguousarray(program["prog_i"], dtype=np.int32)
        pf = np.ascontiguousarray(program["prog_f"], dtype=np.float64)
        program["_keep"] = (pi, pf)
        return (_p(pi), _p(pf), pi.shape[0], int(program["n_slot"]), _p(program["agent_body"]),
                _p(program["agent_obs_len"]), _p(program["store"]), _p(program.get("reward")), _p(program.get("term")),
                _p(program.get("trunc")))

    def __init__(self, model, blob: bytes):
        self.model, self.blob = model, blob
        self.qpos = model.qpos0.copy()
        self.qvel = np.zeros(model.nv)
        self.ctrl = np.zeros(max(model.nu, 1))
        self.warm = np.zeros(model.nv)
        self.sens = np.zeros(max(model.nsensordata, 1))
        self.timestep = np.zeros(1, np.int32)
        self.total = lib().emu_lds_total(blob, len(blob))
        self.dump = np.zeros(self.total)

    def offset(self, name):
        return lib().emu_lds_offset(self.blob, len(self.blob), name.encode())

    def step(self, nsteps=1, skip_frames=1, dbg_stage=0, forward_only=False, actions=None, scatter=None, n_agent=0,
             scatter_mode=0, gather=None, obs=None, program=None, max_steps=1 << 30, reset_warm=None, reset_kind=1,
             reset_sens=None, autoreset=None):
        """``program``: dict(prog_i, prog_f, n_slot, agent_body, agent_obs_len, store, reward, term, trunc) for the
        fused plugin ops.  ``reset_warm``: the first frame starts from the reset image (in-launch reset) with this warm
        start."""
        tags = None if program is None else program.get("tags")
        if tags is not None:                 # list of lists of (kind, id); program["env_base"]: global id of this copy
            num = np.array([len(t) for t in tags] or [0], np.int32)
            adr = np.concatenate([[0], np.cumsum(num)[:-1]]).astype(np.int32)
            ref = np.array([(k << 16) | i for t in tags for k, i in t] or [0], np.int32)
            self._tags = (adr, num, ref)
            lib().emu_set_tags(_p(adr), _p(num), _p(ref), int(program.get("env_base", 0)))
        # (reset_kind 2: the flagged copy is reset without being stepped; autoreset = (flag byte array[1], mode, episode
        # array[1]): the step keeps the flag itself, mjrl_set_autoreset; both need the reset image's sensor readings)
        if reset_sens is not None:
            self._reset_sens = np.ascontiguousarray(reset_sens, dtype=np.float64)
            lib().emu_set_reset_sens(_p(self._reset_sens))
        if reset_warm is not None:
            self._reset_warm = np.ascontiguousarray(reset_warm, dtype=np.float64)
            lib().emu_set_step_reset(int(reset_kind), _p(self._reset_warm))
        if autoreset is not None:
            flag, mode, episode = autoreset
            self._reset_warm = np.ascontiguousarray(self._reset_warm if reset_warm is None else reset_warm, dtype=np.float64)
            lib().emu_set_step_reset(0, _p(self._reset_warm))
            lib().emu_set_autoreset(_p(flag), int(mode), _p(episode))
        else:
            lib().emu_set_autoreset(None, 0, None)
        act_dim = 0 if actions is None else actions.shape[-1]
        obs_dim = 0 if gather is None else gather.shape[-1]
        rc = lib().emu_step(self.blob, len(self.blob), _p(self.qpos), _p(self.qvel), _p(self.ctrl), _p(self.warm),
                            _p(self.sens), _p(self.timestep), _p(actions), _p(scatter), n_agent, act_dim, scatter_mode,
                            _p(gather), obs_dim, _p(obs), skip_frames, nsteps, max_steps, _p(self.dump), dbg_stage,
                            int(forward_only), *self._program_args(program))
        if rc:
            raise RuntimeError(f"emu_step failed with code {rc}")
        return LdsImage(self.dump, self.model, self.offset)
