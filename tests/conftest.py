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


# ---- measured parity errors -------------------------------------------------------------------------------------------------------
# GPU tests report the worst error they measured through the `parity` fixture; the session writes them to gpurun_out/parity.json
# (copied to profiles/parity_rNN.json after a GPU run), so that tolerances can be set from measurements and regressions are visible.
_PARITY = {}


def record_current(key, value):
    """Record a measured error under the running test (usable from helpers that have no fixture access)."""
    name = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]
    d = _PARITY.setdefault(name, {})
    d[key] = max(float(value), d.get(key, 0.0))


@pytest.fixture
def parity(request):
    def rec(key, value):
        d = _PARITY.setdefault(request.node.name, {})
        d[key] = max(float(value), d.get(key, 0.0))
    return rec


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity.json")
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(_PARITY)
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)


def free_port():
    """a TCP port nobody is listening on right now (bind to 0, read it back): rendezvous ports derived from the pid collided now and then with a
    socket of the previous test still in TIME_WAIT"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
