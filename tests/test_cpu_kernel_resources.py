"""Build-time guard for the two VALU-bound sweeps (no GPU needed: hipcc cross-compiles gfx950): the listings of
kernels_surfel.hip / kernels_pose.hip / kernels_pcg.hip, compiled with the Makefile's own flags, must keep the properties
DESIGN.md section 3 and 5 rely on -- 4 wavefronts per SIMD (<= 128 VGPRs), no scratch (spills) in the hot kernels, no
packed-binary32 VALU code (the SLP vectoriser stays off: -13 % when it is on), the gfx950 cross-lane instructions in the
reductions."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "badslam_amd", "csrc")
HIPCC = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)

pytestmark = pytest.mark.skipif(HIPCC is None, reason="needs hipcc (the build container has it)")


def _makefile_flags():
    text = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^HIPFLAGS \?= (.*)$", text, re.M).group(1)
    arch = re.search(r"^ARCH \?= (\S+)", text, re.M).group(1)
    return [f for f in flags.replace("$(ARCH)", arch).split() if f != "-fPIC"]


def _fast_flags(unit):
    """FASTFLAGS + the unit's FASTFTZ_<unit> of the Makefile (the fast arithmetic flavour: ba_launch.h)."""
    text = open(os.path.join(CSRC, "Makefile")).read()
    fast = re.search(r"^FASTFLAGS \?= (.*)$", text, re.M).group(1).split()
    ftz = re.search(r"^FASTFTZ \?= (.*)$", text, re.M).group(1).split()
    per_unit = re.search(r"^FASTFTZ_%s \?=(.*)$" % unit, text, re.M).group(1).strip()
    return fast + (ftz if per_unit == "$(FASTFTZ)" else per_unit.split())


def _compile(d, name, extra, suffix):
    path = str(d / (name + suffix + ".s"))
    subprocess.run([HIPCC] + _makefile_flags() + extra + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", path,
                    os.path.join(CSRC, name + ".hip")], check=True, timeout=900, capture_output=True)
    return open(path).read()


@pytest.fixture(scope="module")
def listings(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa")
    return {name: _compile(d, name, [], "") for name in ("kernels_surfel", "kernels_pose", "kernels_pcg")}


@pytest.fixture(scope="module")
def fast_listings(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa_fast")
    return {name: _compile(d, name, _fast_flags(name), "_fast") for name in ("kernels_surfel", "kernels_pose")}


def _kernels(listing):
    """name -> (body, NumVgprs, ScratchSize, Occupancy)"""
    res = {}
    for m in re.finditer(r"^(_Z\w+):\s*;.*?\n(.*?)s_endpgm.*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)", listing, re.S | re.M):
        res[m.group(1)] = (m.group(2), int(m.group(3)), int(m.group(4)), int(m.group(5)))
    return res


def test_flags_that_parity_and_speed_depend_on_are_in_the_makefile():
    flags = _makefile_flags()
    for needed in ("--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-munsafe-fp-atomics"):
        assert needed in flags, needed
    assert not any("fast-math" in f or f == "-Ofast" for f in flags)


def test_hot_kernels_keep_four_waves_per_simd_without_spills(listings):
    hot = {"kernels_surfel": ("geometry_kernel", "normals_kernel", "activation_kernel"),
           "kernels_pose": ("pose_accumulate_kernel",),
           "kernels_pcg": ("pcg_step1_kernel",)}
    seen = 0
    for unit, prefixes in hot.items():
        for name, (body, vgprs, scratch, occupancy) in _kernels(listings[unit]).items():
            if not any(p in name for p in prefixes):
                continue
            seen += 1
            assert vgprs <= 128 and occupancy >= 4, (name, vgprs, occupancy)
            # (the intrinsics variants of the PCG sweep spill 4 dwords; the alternating sweeps none)
            assert scratch <= (64 if unit == "kernels_pcg" else 0), (name, scratch)
            assert not re.search(r"\bv_pk_(fma|mul|add)_f32\b", body), name       # SLP packing stays off
    assert seen >= 12


def test_fast_flavour_of_the_sweeps_keeps_the_budget_and_is_shorter(listings, fast_listings):
    """The fast arithmetic flavour (Makefile: *_fast.o): its own namespace (no symbol shared with the exact kernels: the linker must
    never pick one flavour's body for the other), the same register budget without spills (FTZ is off in the pose unit for exactly
    that reason), and fewer VALU instructions -- raw v_rcp_f32 / v_sqrt_f32 instead of the correctly rounded sequences."""
    def valu(body):
        return len(re.findall(r"^\s+v_", body, re.M))
    for unit, pick in (("kernels_surfel", "geometry_kernelILb1ELb1ELi1ELb1"), ("kernels_pose", "pose_accumulate_lds_kernelILb1ELb1ELb0")):
        exact = {k: v for k, v in _kernels(listings[unit]).items() if pick in k}
        fast = {k: v for k, v in _kernels(fast_listings[unit]).items() if pick in k}
        assert len(exact) == len(fast) == 1
        (en, (eb, ev, es, eo)), (fn, (fb, fv, fs, fo)) = next(iter(exact.items())), next(iter(fast.items()))
        assert "5exact" in en and "4fast" in fn, (en, fn)
        assert fv <= 128 and fo >= 4 and fs == 0, (fn, fv, fo, fs)
        assert valu(fb) < 0.95 * valu(eb), (unit, valu(fb), valu(eb))
        # the exact flavour's reciprocals end in v_div_fixup_f32 (ba_device.h: rcp_exact), the fast flavour's are bare v_rcp_f32 (what is
        # left there: 1 / baseline_fx of a kernel argument, once per launch)
        # (and a few per-surfel divisions whose operands the compiler cannot bound)
        assert eb.count("v_div_fixup_f32") >= 5 and 4 * fb.count("v_div_fixup_f32") <= eb.count("v_div_fixup_f32"), (unit, eb.count("v_div_fixup_f32"), fb.count("v_div_fixup_f32"))
    # nothing but the sweeps is compiled a second time: no solve / schedule / debug kernel in the fast unit
    names = list(_kernels(fast_listings["kernels_pose"]))
    assert names and all("4fast" in n for n in names), names
    assert not any("pose_solve" in n or "tile_order" in n for n in names)


def test_reductions_use_the_gfx950_cross_lane_instructions(listings):
    kernels = _kernels(listings["kernels_pose"])
    body = next(v[0] for k, v in kernels.items() if "pose_accumulate_kernelILb1ELb1" in k)
    assert "v_permlane32_swap" in body and "v_permlane16_swap" in body and "dpp" in body
    assert "ds_bpermute" not in body                                             # no LDS round trips in the reduction
    assert "global_atomic_add_x2" in body                                        # 64-bit integer atomics on the fixed-point limbs
    assert "global_atomic_add_f32" not in body                                   # ... no float atomics (order-dependent sums)
    assert "global_atomic_cmpswap" not in body                                   # ... and no compare-and-swap loop


def _innermost_loop(body):
    """Lines of the deepest loop of a kernel listing (the per-candidate body of a sweep), from its header label on."""
    lines = body.split("\n")
    depth = max((int(m.group(1)) for m in re.finditer(r"Depth=(\d+)", body)), default=0)
    idx = [i for i, l in enumerate(lines) if f"Depth={depth}" in l]
    return lines[min(idx):max(idx) + 200] if idx else []


def test_pose_sweep_puts_all_five_gathers_in_flight_before_the_first_wait(listings):
    """DESIGN.md section 3, "every gather of a pair in flight at once": in the per-candidate body of the pose sweep the pixel
    word, the cfactor and the three luma footprints are issued (global_load, not flat_load) before the first s_waitcnt on
    vmcnt, and that first wait is a counted one (it lets the later loads stay in flight).  A compiler that sinks one of the
    loads back into the branch that uses it puts the round trips in series again (-7 % measured)."""
    kernels = _kernels(listings["kernels_pose"])
    for form in ("pose_accumulate_kernelILb1ELb1", "pose_accumulate_lds_kernelILb1ELb1"):
        body = next(v[0] for k, v in kernels.items() if form in k)
        assert "flat_load" not in body
        loop = _innermost_loop(body)
        assert loop
        loads_before_wait, first_wait = 0, None
        for line in loop:
            s = line.strip()
            if s.startswith("global_load_dword "):
                loads_before_wait += 1
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", s)
            if m and loads_before_wait:
                first_wait = int(m.group(1))
                break
        assert loads_before_wait == 5, (form, loads_before_wait)
        assert first_wait is not None and first_wait >= 3, (form, first_wait)
        if "lds" in form:
            # persistent form: the tile totals go to the workgroup's table in LDS right behind the reduction (two limbs per
            # total: two ds_add_u64, the compiler's own -- no asm statement, so its waitcnt pass tracks them); global atomics
            # only for drawing tiles and for the flush at the end
            assert len(re.findall(r"\bds_add_u64\b", body)) == 2, (form, body.count("ds_add_u64"))
            assert "ASMSTART\n\tds_add" not in body, form
            assert "global_atomic_add_x2" not in "\n".join(loop[:-200]), form
        else:
            # the atomics of a candidate (two limbs per total) are issued one candidate late: two atomic instructions in the
            # loop, two more after it
            assert body.count("global_atomic_add_x2") == 4, form


def test_sweeps_gather_through_global_loads(listings):
    """Pointers read from the keyframe table are generic to the compiler; load_global() (ba_device.h) says they are global
    memory, so the gathers are global_load (flat_load checks the LDS / scratch apertures and counts on lgkmcnt too)."""
    for unit, prefixes in (("kernels_surfel", ("geometry_kernelILb1ELb1ELi1", "normals_kernel")), ("kernels_pcg", ("pcg_step1_kernel",))):
        for name, (body, *_rest) in _kernels(listings[unit]).items():
            if any(p in name for p in prefixes):
                loop = "\n".join(_innermost_loop(body))
                assert "flat_load" not in loop, name
