import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """The HIP library is built in-tree and git-ignored: build it when a fresh checkout runs the suite (hipcc cross-compiles gfx950
    without a GPU; on the GPU box the prebuilt .so travels with the snapshot)."""
    lib = os.path.join(ROOT, "gpt-st_amd", "lib", "libgptst_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.run([sys.executable, "-m", "gptst_amd.build"], cwd=ROOT, check=True)
