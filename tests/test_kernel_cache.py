"""Model-specialised step kernels: the cache key, the generated size header and the gfx950 cross-build (no GPU)."""
import os

import pytest

from mjrl_amd import blob, kernel_cache, levels, mjcf


def packed(level, **kw):
    return blob.pack(mjcf.compile_mjcf(levels.level_path(level), **kw))


def test_key_follows_the_model_shape_only():
    a = kernel_cache.object_path(kernel_cache.blob_sizes(packed("two_agent.xml")))
    b = kernel_cache.object_path(kernel_cache.blob_sizes(packed("two_agent.xml")))
    c = kernel_cache.object_path(kernel_cache.blob_sizes(packed("two_agent.xml", njmax=40)))
    d = kernel_cache.object_path(kernel_cache.blob_sizes(packed("single_agent.xml")))
    assert a == b and len({a, c, d}) == 3
    assert a.startswith(kernel_cache.CACHE) and a.endswith(".hsaco")


def test_header_defines_every_size_field():
    sizes = kernel_cache.blob_sizes(packed("single_agent.xml"))
    text = kernel_cache.spec_header(sizes)
    for k in blob.SIZE_FIELDS:
        assert f"#define MJRL_SPEC_{k} {sizes[k]}\n" in text
    with pytest.raises(ValueError):
        kernel_cache.blob_sizes(b"\0" * 256)


@pytest.mark.skipif(kernel_cache.hipcc() is None, reason="no hipcc in this image")
def test_cross_build_produces_a_code_object_with_the_entry_point():
    path = kernel_cache.code_object(packed("sensor_touch.xml"))
    assert path and os.path.getsize(path) > 10000
    data = open(path, "rb").read()
    assert b"mjrl_step_kernel_spec" in data and b"mjrl_spec_sizes" in data
    assert kernel_cache.code_object(packed("sensor_touch.xml"), build=False) == path


@pytest.mark.skipif(kernel_cache.hipcc() is None, reason="no hipcc in this image")
def test_ranks_that_build_the_same_shape_at_once_all_end_with_a_valid_object(tmp_path, monkeypatch):
    """The ranks of a multi-GPU run find the cache cold together: each builds the code object in a directory of its own
    and moves it into place atomically, so every rank ends with the same complete file whoever wins (what the N > 1
    launch of bench.py relies on; one box, several processes here)."""
    import multiprocessing as mp
    import shutil
    cache = tmp_path / "_spec"
    blob_bytes = packed("sensor_framexaxis.xml", njmax=24)        # a shape nobody has cached
    ctx = mp.get_context("fork")
    queue = ctx.Queue()

    def rank(k):
        kernel_cache.CACHE = str(cache)
        queue.put((k, kernel_cache.code_object(blob_bytes)))

    procs = [ctx.Process(target=rank, args=(k,)) for k in range(3)]
    for p in procs:
        p.start()
    paths = dict(queue.get(timeout=600) for _ in procs)
    for p in procs:
        p.join()
        assert p.exitcode == 0
    assert len(set(paths.values())) == 1
    path = next(iter(paths.values()))
    data = open(path, "rb").read()
    assert len(data) > 10000 and b"mjrl_step_kernel_spec" in data
    assert sorted(f for f in os.listdir(cache) if not f.startswith("tmp")) == [os.path.basename(path)]
    shutil.rmtree(cache)
