"""Model-specialised step kernels.

The generic ``mjrl_step_kernel`` in libmjrl_hip.so takes every size of the model at run time, so each wave spends a
good part of a step on address arithmetic and keeps the LDS layout in (spilled) scalar registers.  For a given model
shape the same source is rebuilt with the sizes as compile-time constants (``csrc/mjrl_spec_kernel.hip``,
``hipcc --genco``) and attached with ``mjrl_load_kernel``.  Code objects are cached in-tree under ``csrc/_spec/``,
keyed by the sizes and a digest of the kernel sources, so a prebuilt cache travels with the tree.

The reference has nothing like this (MuJoCo's C step is shape-generic); it is the stand-in for a tracing compiler.
"""
from __future__ import annotations

import glob
import hashlib
import os
import shutil
import struct
import subprocess
import tempfile

from . import blob as _blob

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
CACHE = os.path.join(CSRC, "_spec")
SOURCES = ["mjrl_spec_kernel.hip", "mjrl_step.h", "mjrl_collide.h", "mjrl_math.h", "mjrl_wave.h", "mjrl_model.h",
           "mjrl_layout.h"]
# (max-ilp: the scheduler orders for instruction-level parallelism instead of for occupancy -- the kernel's occupancy is
# fixed by its LDS image and its register cap anyway, and a wave waits on its own dependent chains: +0.5 % on the full
# batches, +1.4 % on a batch of one wave per SIMD, profiles/r04_scheduling_flag_variants.txt; no change in any result)
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-Wno-unused-value",
         "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
if os.environ.get("MJRL_SPEC_FLAGS"):          # experiments only: extra compiler flags for the specialised kernel
    FLAGS = FLAGS + os.environ["MJRL_SPEC_FLAGS"].split()


def blob_sizes(blob: bytes) -> dict:
    n = len(_blob.SIZE_FIELDS)
    magic, version = struct.unpack_from("<ii", blob, 0)
    if magic != _blob.MAGIC or version != _blob.VERSION:
        raise ValueError("not a model blob of this layout version")
    return dict(zip(_blob.SIZE_FIELDS, struct.unpack_from(f"<{n}i", blob, 8)))


def _source_digest() -> str:
    h = hashlib.sha1()
    for name in SOURCES:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def _flags_digest(extra=()) -> str:
    """Builds with other flags (the diagnostic build, ``MJRL_SPEC_FLAGS=-DMJRL_DIAG``; the few-copies build) are cached
    side by side."""
    return hashlib.sha1(" ".join([*FLAGS, *extra]).encode()).hexdigest()[:6]


def spec_header(sizes: dict) -> str:
    return "".join(f"#define MJRL_SPEC_{k} {int(sizes[k])}\n" for k in _blob.SIZE_FIELDS)


FEW_FLAGS = ("-DMJRL_FEW=1",)       # the build for batches of at most one wave per SIMD (csrc/mjrl_step.h stage_pgs)


def object_path(sizes: dict, few: bool = False) -> str:
    key = hashlib.sha1(spec_header(sizes).encode()).hexdigest()[:16]
    return os.path.join(CACHE, f"step_{key}{_flags_digest(FEW_FLAGS if few else ())}_{_source_digest()}.hsaco")


def hipcc() -> str | None:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def code_object(blob: bytes, build: bool = True, few: bool = False) -> str | None:
    """Path of the code object for this blob's shape; built on a cache miss when hipcc is present, else None.
    ``few``: the build for a batch that leaves every SIMD at most one wave (``mjrl_size(h, "few")``): its solver forms may
    use the whole register file."""
    sizes = blob_sizes(blob)
    if os.environ.get("MJRL_SPEC_OBJECT"):       # experiments only: A/B a saved code object of the same model shape
        return os.environ["MJRL_SPEC_OBJECT"]
    path = object_path(sizes, few)
    if os.path.exists(path):
        return path
    cc = hipcc()
    if not build or cc is None:
        return None
    try:
        os.makedirs(CACHE, exist_ok=True)
        probe = tempfile.TemporaryDirectory(dir=CACHE)
    except OSError:
        return None                 # read-only install: the generic kernel of libmjrl_hip.so stays in place
    with probe as tmp:
        hdr = os.path.join(tmp, "spec.h")
        with open(hdr, "w") as f:
            f.write(spec_header(sizes))
        out = os.path.join(tmp, "step.hsaco")
        # (the digest of the kernel sources goes into the object: mjrl_load_kernel compares it with the library's)
        cmd = [cc, "--genco", *FLAGS, *(FEW_FLAGS if few else ()), f'-DMJRL_SPEC_HEADER="{hdr}"',
               f"-DMJRL_SOURCE_DIGEST=0x{_source_digest()}ull", "-I", CSRC,
               os.path.join(CSRC, "mjrl_spec_kernel.hip"), "-o", out]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"specialised kernel build failed:\n{' '.join(cmd)}\n{res.stderr[-4000:]}")
        os.replace(out, path)     # atomic: concurrent ranks race benignly
    shape = os.path.basename(path).rsplit("_", 1)[0]
    for stale in glob.glob(os.path.join(CACHE, shape + "_*.hsaco")):     # same shape, older kernel sources
        if stale != path:
            try:
                os.remove(stale)
            except OSError:
                pass
    return path
