"""CPU-only checks of the drop-in boundary: the C-ABI libraries load, export every symbol their
headers declare, and refuse to work without a GPU (no silent CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)) - {prefix + "allreduce_fn"})


def test_hip_library_exports_every_declared_symbol():
    from badslam_amd import capi
    lib = ctypes.CDLL(capi.LIB_PATH)
    names = _declared("badslam_hip.h", "bahip_")
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the Python prototypes cover the same set
    assert sorted(capi.SIGNATURES) == names


def test_host_library_exports_every_declared_symbol():
    from badslam_amd import directba
    lib = ctypes.CDLL(directba.HOST_LIB_PATH)
    names = _declared("badslam_directba.h", "dba_")
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from badslam_amd import capi, lowlevel
    assert capi.load().bahip_device_count() == 0
    with pytest.raises(capi.BackendError):
        lowlevel.Context()
    from badslam_amd.directba import DirectBA
    with pytest.raises(capi.BackendError):
        DirectBA(1000, 1 / 5000, 40.0, 2, 64, 48, [24, 24, 31.5, 23.5], [24, 24, 31.5, 23.5])


def test_product_package_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    offenders = []
    for top in ("badslam_amd", "scripts", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".sh", ".h", ".hip", ".cc", ".c")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"(from\s+oracle|import\s+oracle|liboracle|build_oracle|oracle\.binding)", text) or \
                       (top != "scripts" and "oracle/" in text):
                        offenders.append(os.path.relpath(os.path.join(dirpath, f), ROOT))
    assert not offenders, offenders


def test_build_refuses_an_unpinned_compiler(tmp_path, monkeypatch):
    """__graft_entry__.check_toolchain: the pinned `hipcc --version` passes, any other fails the build loudly
    unless BADSLAM_ACCEPT_TOOLCHAIN=1 says the GPU suite is about to be re-run with it."""
    import shutil
    import __graft_entry__ as entry
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this box")
    entry.check_toolchain()
    other = tmp_path / "TOOLCHAIN.txt"
    other.write_text("HIP version: 0.0.0-none\nAMD clang version 0.0.0\n")
    monkeypatch.delenv("BADSLAM_ACCEPT_TOOLCHAIN", raising=False)
    with pytest.raises(RuntimeError, match="not the pinned toolchain"):
        entry.check_toolchain(str(other))
    monkeypatch.setenv("BADSLAM_ACCEPT_TOOLCHAIN", "1")
    entry.check_toolchain(str(other))
