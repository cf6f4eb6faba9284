"""TEST INFRASTRUCTURE -- pytest plugin: `python -m pytest -p host_exec.pytest_hostexec -m gpu tests/test_gpu_parity.py -k ...` runs the
selected `-m gpu` parity tests against the host-executed kernels (tests/host_exec/harness.hpp) instead of a GPU: the whole session is
wrapped in hostmode.host_mode().  Used by tests/test_kernels_on_host.py for a curated small-shape subset, and by hand for anything else."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

_ctx = None


def pytest_sessionstart(session):
    global _ctx
    import build_host
    import hostmode
    lib = build_host.build(os.environ.get("HOSTEXEC_DIR", "/tmp/rnnpose_hostexec"))
    _ctx = hostmode.host_mode(lib)
    _ctx.__enter__()
    from rnnpose_amd import build
    build.build = lambda *a, **k: lib           # the GPU tests' fixtures call build.build(): there is nothing to build for a GPU here


def pytest_sessionfinish(session, exitstatus):
    global _ctx
    if _ctx is not None:
        _ctx.__exit__(None, None, None)
        _ctx = None
