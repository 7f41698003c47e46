import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an AMD GPU (runs the HIP engine through the C ABI)")
    lib = os.path.join(ROOT, "pyaudiolocalization_amd", "libpal_hip.so")
    if not os.path.exists(lib):          # fresh checkout: build once (hipcc cross-compiles without a GPU)
        subprocess.run(["make", "-C", os.path.join(ROOT, "pyaudiolocalization_amd", "csrc"), "-j", "4"], check=True)


def pytest_sessionstart(session):
    """Start multiprocessing's fork server NOW, before any test touches the GPU.  The two-rank GPU test
    (tests/test_gpu_stream.py) starts its rank processes through it: they are forks of this clean server, so no process
    that has initialised HIP ever forks or execs (the GPU boxes forbid an exec from such a process)."""
    import multiprocessing
    import multiprocessing.forkserver as forkserver
    global FORKSERVER_OK
    try:
        multiprocessing.get_context("forkserver")
        forkserver.ensure_running()
        FORKSERVER_OK = True
    except Exception as exc:                     # reported, never fatal: only the two-rank GPU tests need it
        FORKSERVER_OK = False
        print(f"[conftest] fork server not started: {exc}")


FORKSERVER_OK = False


def require_forkserver():
    """Skip (never respawn): a test that starts processes through the fork server must not let multiprocessing relaunch
    it now - that would be a fork + exec from a process that has initialised HIP, which the GPU boxes forbid."""
    import multiprocessing.forkserver as forkserver
    alive = False
    try:
        pid = getattr(forkserver._forkserver, "_forkserver_pid", None)
        if pid:
            os.kill(pid, 0)
            alive = True
    except Exception:
        alive = False
    if not (FORKSERVER_OK and alive):
        pytest.skip("the fork server of the test session is not running (it must predate every HIP call)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def engine():
    """One HIP engine for the GPU tests; creation fails loudly when no GPU is visible."""
    from pyaudiolocalization_amd import Engine
    eng = Engine(0)
    yield eng
    eng.close()
