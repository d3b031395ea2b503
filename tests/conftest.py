import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the recommended setting of INTEGRATION.md for processes that use the batch entry points; the test run is such an application
# (the package itself never edits the environment)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """Both shared libraries are build artefacts (git-ignored): build them when a fresh checkout lacks them
    (hipcc cross-compiles gfx950 without a GPU; gcc builds the oracle)."""
    import subprocess
    lib = os.path.join(ROOT, "3d_reconstruction_project_amd", "lib", "libr3d_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["bash", os.path.join(ROOT, "3d_reconstruction_project_amd", "csrc", "build.sh")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libr3d_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def r3d():
    """The product package (its directory name starts with a digit, so it is imported by string)."""
    _ensure_built()
    return importlib.import_module("3d_reconstruction_project_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("3d_reconstruction_project_amd.synth")


GOLDEN = os.path.join(ROOT, "tests", "golden")
