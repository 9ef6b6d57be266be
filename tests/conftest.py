import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

entry.load_package()

# A handle for a batch that leaves every SIMD at most one wave picks the step kernel's build with the big solver forms
# (mjrl_size "few").  The tests' batches are all that small, and most of them are there for the build the headline runs:
# they get it unless a test asks for the other kind (monkeypatch.setenv("MJRL_FEW", "1")).
os.environ.setdefault("MJRL_FEW", "0")

LEVELS = os.path.join(ROOT, "tests", "levels")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
