import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# torch bundles its own HIP runtime (torch/lib/libamdhip64.so, ROCm 7.0); the backend library resolves libamdhip64 through the
# system's (/opt/rocm, 7.2).  Whichever is loaded first serves the whole process, and torch does not find its GPUs on the
# system's copy ("No HIP GPUs are available") -- so a process that uses both (the loopback and gloo tests, bench.py) imports
# torch FIRST.  A full run did that by accident (collection imports tests/test_cpu_multigpu_gloo.py); a partial run did not.
# (BADSLAM_TESTS_NO_TORCH=1 skips it: for a quick run of tests that use the backend through ctypes only.)
try:
    if not os.environ.get("BADSLAM_TESTS_NO_TORCH"):
        import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test taking more than ~20 s")


def pytest_sessionstart(session):
    """The libraries are build products (not in git).  If a checkout is tested before __graft_entry__.build() ran and a
    compiler is at hand (the build container), build them; on a box without hipcc the tests fail with their own messages."""
    needed = [os.path.join(ROOT, "badslam_amd", "lib", n) for n in ("libbadslam_hip.so", "libbadslam_host.so", "test_directba", "ba_tum", "test_route_b")]
    needed.append(os.path.join(ROOT, "oracle", "liboracle.so"))
    if all(os.path.exists(p) for p in needed):
        return
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        return
    import __graft_entry__
    __graft_entry__.build()
