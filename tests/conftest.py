import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

entry.load_package()

# A handle for a batch that leaves every SIMD at most one wave picks the step kernel's build with the big solver forms
# (mjrl_size "few": what a reference-style numEnvs=1 env, smoke() and bench.py's configs 2 and 5 run); larger batches run
# the full-batch build (the headline).  The tests' batches are all small, so the kind is chosen here: a test that takes
# the `few_build` fixture runs once with each build; every other test runs the full-batch build.
os.environ.setdefault("MJRL_FEW", "0")

LEVELS = os.path.join(ROOT, "tests", "levels")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_generate_tests(metafunc):
    for name in ("few_build", "emu_few"):
        if name in metafunc.fixturenames:
            metafunc.parametrize(name, ["full", "few"], indirect=True)


@pytest.fixture
def few_build(request, monkeypatch):
    """Handles created by the test attach the full-batch build ("full") or the build for batches of at most one wave
    per SIMD ("few": -DMJRL_FEW=1 specialised kernel / StepArgs::few of the generic ones, no longest-first dispatch)."""
    kind = getattr(request, "param", "full")
    monkeypatch.setenv("MJRL_FEW", "1" if kind == "few" else "0")
    return kind


@pytest.fixture
def emu_few(request):
    """The same choice for the CPU emulation of the device source (StepArgs::few of the emulated waves)."""
    from tests.emu import emu
    kind = getattr(request, "param", "full")
    emu.lib().emu_set_few(1 if kind == "few" else 0)
    yield kind
    emu.lib().emu_set_few(0)


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def levels_dir():
    return LEVELS
