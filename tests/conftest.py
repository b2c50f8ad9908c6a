import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """GPU sessions: the two ranks of tests/test_gpu_multiprocess.py are started here, before any test (and so before
    this process initialises the GPU: from then on it must not start other programs).  They run beside the session."""
    import socket
    import subprocess
    import tempfile
    expr = session.config.getoption("markexpr") or ""
    if "not gpu" in expr or not os.path.exists("/dev/kfd") or session.config.getoption("collectonly"):
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = tempfile.mkdtemp(prefix="p4s_mp_")
    log = os.path.join(d, "ranks.log")
    world = 2
    with open(log, "w") as f:
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_shard_worker.py"), str(r), str(world),
                                   str(port), d], stdout=f, stderr=subprocess.STDOUT, cwd=ROOT) for r in range(world)]
    session.config._p4s_mp_job = {"procs": procs, "dir": d, "log": log, "world": world}


def pytest_sessionfinish(session, exitstatus):
    job = getattr(session.config, "_p4s_mp_job", None)
    if job:
        for p in job["procs"]:
            if p.poll() is None:
                p.kill()


def load_vectors(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def reference_vectors():
    return load_vectors("reference_vectors.json")


@pytest.fixture(scope="session")
def l0_vectors():
    return load_vectors("l0_ref_vectors.json")
