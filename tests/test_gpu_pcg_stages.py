"""The PCG scheme driven stage by stage through the C ABI (bahip_pcg_begin / init / init2 / step1 / step2 / step3 / update_*:
one entry point per *CUDA function of B/kernels.h:397-491), with the call sequence of the reference's own driver
(B/direct_ba_pcg.cc:229-646: per-keyframe PCGInit and PCGStep1 calls, host-side stopping rule) -- against
bahip_pcg_iteration, which runs the keyframe loops as single sweeps and the stopping rule on the device: the same bits, vectors,
inner step count and final state (and therefore the oracle's, tests/test_gpu_intrinsics_pcg_vs_oracle.py)."""
import ctypes as C

import numpy as np
import pytest

from badslam_amd import capi, lowlevel as ll
from oracle import binding as ob
from tests import common
from tests.test_gpu_intrinsics_pcg_vs_oracle import _pcg_setup

pytestmark = pytest.mark.gpu

INVALID = 0xFFFFFFFF


class _Vec:
    """A 1 x n float device buffer."""

    def __init__(self, ctx, n):
        self.buf = ll.DeviceBuffer2D(ctx, 1, max(1, n), np.float32).clear(0)
        self.n = n

    @property
    def ptr(self):
        return self.buf.ptr

    def get(self):
        return self.buf.download()[0, :self.n].copy()


def _pose_index(k, gauge):
    return INVALID if k == gauge else 6 * (k if k < gauge else k - 1)


@pytest.mark.parametrize("mode", ["poses+geometry", "all"])
def test_stage_api_reproduces_the_fused_iteration(mode):
    scene = common.small_scene(num_keyframes=5, seed=21)
    di = ci = (mode == "all")
    gauge = 1
    # the fused call on one scene ...
    _, g, data, _ = _pcg_setup(scene, mode)
    g.update_surfel_normals()
    steps_ref, _ = g.pcg_iteration(optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=di, optimize_color_intrinsics=ci,
                                   gauge_keyframe=gauge)
    K, N = len(g.keyframes), data.shape[1]
    S = g.cf_w * g.cf_h
    U = 6 * (K - 1) + 3 * N + ((5 + S + 4) if di else 0)
    delta_ref = g.read_pcg_vector(2, U)
    surfels_ref = g.download_surfels()
    cf_ref = g.cfactor.download()

    # ... and the reference's driver, stage by stage, on an identical one
    _, h, data2, _ = _pcg_setup(scene, mode)
    assert np.array_equal(data, data2)
    h.update_surfel_normals()
    lib, ctx = h.ctx.lib, h.ctx.handle
    layout = capi.PCGLayout(1, 1, int(di), int(ci), 1, 1, U, 6 * (K - 1), (6 * (K - 1) + 3 * N) if di else INVALID,
                            (6 * (K - 1) + 3 * N + 5 + S) if ci else INVALID)
    r, M, delta, gv, p = (_Vec(h.ctx, U) for _ in range(5))
    an, ad, bn = (_Vec(h.ctx, 1) for _ in range(3))
    s = h.surfels_struct()
    frames = [h.frame_struct(k) for k in range(K)]
    # frame_T_global of a pose with the bits the backend's own keyframe table holds (the oracle's SE(3) inverse is the device's)
    Fs = [(C.c_float * 12)(*[float(v) for v in ob.se3_matrix3x4(ob.se3_inverse(ob.SE3.from_array(h.keyframes[k]["pose"])))]) for k in range(K)]
    if di:
        # a smaller layout first on the same context: the exact accumulators then REGROW for the real layout (its
        # head has the S cfactor cells), and the 64-byte control block -- an allocation of its own -- must survive that
        small = capi.PCGLayout(1, 1, 0, 0, 1, 1, 6 * (K - 1) + 3 * N, 6 * (K - 1), INVALID, INVALID)
        capi.check(lib.bahip_pcg_begin(ctx, C.byref(small), N))
    capi.check(lib.bahip_pcg_begin(ctx, C.byref(layout), N))
    for k in range(K):
        capi.check(lib.bahip_pcg_init(ctx, C.byref(layout), C.byref(frames[k]), Fs[k], _pose_index(k, gauge), int(k != gauge), C.byref(s), r.ptr, M.ptr))
    capi.check(lib.bahip_pcg_init2(ctx, C.byref(layout), N, h.dp.a, r.ptr, M.ptr, delta.ptr, gv.ptr, p.ptr, an.ptr))
    # r and M as the fused call assembled them (max_inner_iterations = 0 on a third scene would repeat the work: compare below
    # through delta, which depends on every entry of both)
    prev, no_improvement, steps = np.inf, 0, 0
    for step in range(30):
        steps += 1
        if step > 0:
            an, bn = bn, an                                   # B/direct_ba_pcg.cc:388-393
            gv.buf.clear(0)
        for k in range(K):
            capi.check(lib.bahip_pcg_step1(ctx, C.byref(layout), C.byref(frames[k]), Fs[k], _pose_index(k, gauge), int(k != gauge), C.byref(s), p.ptr, gv.ptr))
        capi.check(lib.bahip_pcg_step2(ctx, C.byref(layout), N, r.ptr, M.ptr, delta.ptr, gv.ptr, p.ptr, an.ptr, ad.ptr, bn.ptr))
        r_norm = float(np.sqrt(np.float32(bn.get()[0])))
        if r_norm < prev - 1e-3:
            no_improvement = 0
        else:
            no_improvement += 1
            if no_improvement >= 3:
                break
        prev = r_norm
        if step < 29:
            capi.check(lib.bahip_pcg_step3(ctx, C.byref(layout), N, gv.ptr, p.ptr, an.ptr, bn.ptr))
    assert steps == steps_ref, (steps, steps_ref)
    assert np.array_equal(delta.get().view(np.uint32), delta_ref.view(np.uint32)), np.abs(delta.get() - delta_ref).max()
    capi.check(lib.bahip_update_surfels_from_pcg_delta(ctx, C.byref(s), 1, 6 * (K - 1), delta.ptr))
    assert np.array_equal(h.download_surfels()[:8].view(np.uint32), surfels_ref[:8].view(np.uint32))
    if di:
        capi.check(lib.bahip_update_cfactors_from_pcg_delta(ctx, 6 * (K - 1) + 3 * N + 5, delta.ptr))
        assert np.array_equal(h.cfactor.download().view(np.uint32), cf_ref.view(np.uint32))
