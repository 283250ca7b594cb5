#!/usr/bin/env python3
"""Generates tests/golden/jacobians.json from the reference's own derivation script.

applications/badslam/scripts/jacobians_derivation.py defines, in sympy, the residuals of the direct BA and the functions
they are composed of (SE3exp, SE3Inverse, Project, Unproject, CorrectDepth, InterpolateBilinear, ...) and derives their
Jacobians symbolically (it prints C++ for them).  Here the script is IMPORTED from /root/reference (this only works in the
build container; the GPU box has no /root/reference, which is why the vectors are committed), the same function chains are
evaluated numerically with 60-digit floats, and the Jacobians are taken by central differences with a step of 1e-20 - exact
to all printed digits, and free of the 0/0 the symbolic rotation derivative runs into at omega = 0 under current sympy
(SURVEY 8c).  Each case stores the inputs and the Jacobian; tests/test_cpu_golden_jacobians.py feeds the inputs to the
oracle's Jacobian functions (oracle_internal.h: jac_*).

Run:  python scripts/make_golden_jacobians.py      (needs /root/reference and sympy)
"""
import json
import os
import sys

import numpy as np
import sympy
import sympy.printing.cxx as _cxx

sys.modules.setdefault("sympy.printing.cxxcode", _cxx)       # sympy >= 1.10 renamed the module the script imports
REF_SCRIPTS = "/root/reference/applications/badslam/scripts"
sys.path.insert(0, REF_SCRIPTS)
import jacobians_derivation as ref                             # noqa: E402  (the reference, imported - not copied)

PREC = 60
H = sympy.Float("1e-20", PREC)
# the script models frac() as an unevaluated function with derivative 1; numerically it is x - floor(x)
ref.frac = lambda v: v - sympy.floor(v)


def F(x):
    return sympy.Float(repr(float(x)), PREC)


def col(values):
    return sympy.Matrix([[F(v)] for v in values])


def mat34(m):
    return sympy.Matrix(3, 4, [F(v) for v in np.asarray(m).reshape(-1)])


def central_difference(f, x0):
    """f: list of sympy numbers -> sympy number.  Returns the gradient at x0 (list of floats)."""
    out = []
    for i in range(len(x0)):
        xp = list(x0); xm = list(x0)
        xp[i] = xp[i] + H
        xm[i] = xm[i] - H
        out.append(float((f(xp) - f(xm)) / (2 * H)))
    return out


def random_rigid(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    t = rng.uniform(-1, 1, 3)
    return np.hstack([R, t[:, None]])


def main():
    rng = np.random.default_rng(20240924)
    cases = {k: [] for k in ("depth_pose", "depth_surfel", "depth_intrinsics", "depth_correction", "descriptor_pose",
                             "descriptor_surfel", "descriptor_color_intrinsics")}
    zero6 = [sympy.Float(0, PREC)] * 6
    for _ in range(6):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        G = random_rigid(rng)                                   # global_T_frame
        l = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.6, 0.6), rng.uniform(1.0, 3.0)])   # local measured point
        s = (G[:, :3] @ l + G[:, 3]) + rng.normal(scale=0.01, size=3)                           # surfel near it
        n_s, G_s, l_s, s_s = col(n), mat34(G), col(l), col(s)

        # --- depth residual wrt the pose delta: jacobians_derivation.py:206-214 ---
        def depth_pose(T):
            M = ref.SE3exp(sympy.Matrix(T))
            p = ref.MatrixVectorMultiplyHomogeneous(G_s, ref.MatrixVectorMultiplyHomogeneous(M, l_s))
            return ref.DotProduct3(n_s, p - s_s)
        cases["depth_pose"].append(dict(surfel_normal=n.tolist(), global_T_frame=G.reshape(-1).tolist(), local_point=l.tolist(),
                                        surfel_pos=s.tolist(), jacobian=central_difference(depth_pose, zero6)))

        # --- depth residual wrt the surfel offset t: :222-228 ---
        g_s = ref.MatrixVectorMultiplyHomogeneous(G_s, l_s)
        def depth_surfel(t):
            return ref.DotProduct3(n_s, g_s - (s_s + t[0] * n_s))
        cases["depth_surfel"].append(dict(surfel_normal=n.tolist(), jacobian=central_difference(depth_surfel, [sympy.Float(0, PREC)])))

        # --- depth residual wrt (fx_inv, fy_inv, cx_inv, cy_inv): :233-240 ---
        x, y = int(rng.integers(0, 640)), int(rng.integers(0, 480))
        depth = float(rng.uniform(0.8, 3.0))
        intr = [1 / 525.0, 1 / 520.0, -319.5 / 525.0, -239.5 / 520.0]
        def depth_intr(v):
            p = ref.Unproject(F(x), F(y), F(depth), v[0], v[1], v[2], v[3])
            return ref.DotProduct3(n_s, ref.MatrixVectorMultiplyHomogeneous(G_s, p) - s_s)
        cases["depth_intrinsics"].append(dict(surfel_normal=n.tolist(), global_T_frame=G.reshape(-1).tolist(), x=x, y=y, depth=depth,
                                              intrinsics=intr, jacobian=central_difference(depth_intr, [F(v) for v in intr])))

        # --- depth residual wrt (cfactor, a): :245-253 ---
        cfactor, a = float(rng.uniform(-0.02, 0.02)), float(rng.uniform(-0.1, 0.3))
        raw_inv_depth = float(1.0 / rng.uniform(0.8, 3.0))
        def depth_corr(v):
            d = ref.CorrectDepth(v[0], v[1], F(raw_inv_depth))
            p = ref.Unproject(F(x), F(y), d, *[F(q) for q in intr])
            return ref.DotProduct3(n_s, ref.MatrixVectorMultiplyHomogeneous(G_s, p) - s_s)
        cases["depth_correction"].append(dict(surfel_normal=n.tolist(), global_T_frame=G.reshape(-1).tolist(), x=x, y=y, intrinsics=intr,
                                              cfactor=cfactor, a=a, raw_inv_depth=raw_inv_depth,
                                              jacobian=central_difference(depth_corr, [F(cfactor), F(a)])))

        # --- descriptor-type residual (one bilinear lookup) ---
        fx, fy, cx, cy = 525.0, 520.0, 320.0, 240.0
        ls = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.4, 0.4), rng.uniform(1.0, 3.0)])   # surfel in the keyframe frame
        texels = rng.uniform(0, 1, 4)                           # top_left, top_right, bottom_left, bottom_right
        tl, tr, bl, br = [F(v) for v in texels]
        ls_s = col(ls)
        def lookup(p):
            return ref.InterpolateBilinear(p[0], p[1], tl, tr, bl, br)
        # wrt the pose delta: :268-277
        def desc_pose(T):
            M = ref.SE3Inverse(ref.SE3exp(sympy.Matrix(T)))
            return lookup(ref.Project(ref.MatrixVectorMultiplyHomogeneous(M, ls_s), F(fx), F(fy), F(cx), F(cy)))
        cases["descriptor_pose"].append(dict(local_surfel_pos=ls.tolist(), texels=texels.tolist(), camera=[fx, fy, cx, cy],
                                             jacobian=central_difference(desc_pose, zero6)))
        # wrt the surfel offset t: :285-293 (frame_T_global F, global surfel s2 with F * s2 = ls)
        Fm = random_rigid(rng)
        s2 = Fm[:, :3].T @ (ls - Fm[:, 3])
        F_s, s2_s = mat34(Fm), col(s2)
        def desc_surfel(t):
            p = ref.MatrixVectorMultiplyHomogeneous(F_s, s2_s + t[0] * n_s)
            return lookup(ref.Project(p, F(fx), F(fy), F(cx), F(cy)))
        cases["descriptor_surfel"].append(dict(surfel_normal=n.tolist(), frame_T_global=Fm.reshape(-1).tolist(), surfel_pos=s2.tolist(),
                                               texels=texels.tolist(), camera=[fx, fy, cx, cy],
                                               jacobian=central_difference(desc_surfel, [sympy.Float(0, PREC)])))
        # wrt (fx, fy, cx, cy) of the colour camera, the parametrisation B/kernel_opt_intrinsics.cu:176-199 uses
        def desc_color(v):
            return lookup(ref.Project(ls_s, v[0], v[1], v[2], v[3]))
        cases["descriptor_color_intrinsics"].append(dict(local_surfel_pos=ls.tolist(), texels=texels.tolist(), camera=[fx, fy, cx, cy],
                                                         jacobian=central_difference(desc_color, [F(fx), F(fy), F(cx), F(cy)])))

    out = dict(source="applications/badslam/scripts/jacobians_derivation.py (imported), evaluated with %d-digit floats, central differences h = 1e-20" % PREC,
               generator="scripts/make_golden_jacobians.py", sympy=sympy.__version__, cases=cases)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "jacobians.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, {k: len(v) for k, v in cases.items()})


if __name__ == "__main__":
    main()
