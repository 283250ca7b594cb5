"""Route B of INTEGRATION.md without a GPU: the shim (badslam_amd/host/route_b/kernels_hip.cc) must implement every function its
stand-in for the reference's B/kernels.h declares, with the reference's argument lists, and link against the two libraries."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RB = os.path.join(ROOT, "badslam_amd", "host", "route_b")


def _functions(text):
    return set(re.findall(r"^void (\w+CUDA)\(", text, re.M))


def test_shim_defines_every_declared_reference_function():
    header = open(os.path.join(RB, "badslam", "kernels.h")).read()
    shim = open(os.path.join(RB, "kernels_hip.cc")).read()
    declared, defined = _functions(header), _functions(shim)
    assert len(declared) >= 11 and declared == defined, (declared - defined, defined - declared)
    # every function ends in exactly one call of the C ABI (plus the binding calls)
    for name in declared:
        body = shim[shim.index(f"void {name}("):]
        body = body[:body.index("\n}\n") + 3]
        assert re.search(r"BAHIP_CHECKED_CALL\(bahip_(?!set_|context_)", body), name


def test_signatures_follow_the_reference_header():
    """Where the reference tree is at hand (the build container), the parameter NAMES of every shim function are compared
    with B/kernels.h, in order; elsewhere the committed stand-in is what documents them."""
    ref = "/root/reference/applications/badslam/src/badslam/kernels.h"
    if not os.path.exists(ref):
        import pytest
        pytest.skip("reference tree not present")
    ref_text, mine = open(ref).read(), open(os.path.join(RB, "badslam", "kernels.h")).read()

    def params(text, name):
        m = re.search(r"void " + name + r"\((.*?)\);", text, re.S)
        assert m, name
        names = []
        for p in re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S).split(","):
            p = re.sub(r"=.*", "", p).strip()
            names.append(re.findall(r"(\w+)\s*$", p)[0])
        return names

    for name in sorted(_functions(mine)):
        assert params(mine, name) == params(ref_text, name), name


def test_route_b_binary_was_linked():
    binary = os.path.join(ROOT, "badslam_amd", "lib", "test_route_b")
    assert os.path.exists(binary)
    out = subprocess.run(["ldd", binary], capture_output=True, text=True).stdout
    assert "libbadslam_host.so" in out and "libbadslam_hip.so" in out
