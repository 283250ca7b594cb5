#!/usr/bin/env python3
"""Generates tests/golden/functions.json: VALUES of the building blocks the reference's derivation script defines
(applications/badslam/scripts/jacobians_derivation.py: CorrectDepth, Project, Unproject, InterpolateBilinear, SO3exp +
QuaternionToRotationMatrix), evaluated with 60-digit floats on random inputs.  The script is IMPORTED from /root/reference
(build container only; the vectors are committed because the GPU box has no /root/reference).
tests/test_cpu_golden_functions.py feeds the inputs to the oracle's own functions: this pins the association arithmetic of
the oracle -- depth calibration, projection, unprojection, the bilinear sampler's weights, the exponential map -- to the
reference's formulas, next to the Jacobians (scripts/make_golden_jacobians.py).

Run:  python scripts/make_golden_functions.py      (needs /root/reference and sympy)
"""
import json
import os
import sys

import numpy as np
import sympy
import sympy.printing.cxx as _cxx

sys.modules.setdefault("sympy.printing.cxxcode", _cxx)       # sympy >= 1.10 renamed the module the script imports
sys.path.insert(0, "/root/reference/applications/badslam/scripts")
import jacobians_derivation as ref                             # noqa: E402  (the reference, imported - not copied)

PREC = 60
ref.frac = lambda v: v - sympy.floor(v)                        # the script leaves frac() unevaluated


def F(x):
    return sympy.Float(repr(float(x)), PREC)


def f32(x):
    return float(np.float32(x))


def main():
    rng = np.random.default_rng(20260924)
    out = {"correct_depth": [], "project": [], "unproject": [], "bilinear": [], "so3_exp": []}
    for _ in range(16):
        cfactor, a = f32(rng.uniform(-0.02, 0.02)), f32(rng.uniform(-0.05, 0.05))
        raw, scale = int(rng.integers(500, 40000)), f32(1.0 / 5000.0)
        inv_depth = 1 / (F(scale) * raw)
        out["correct_depth"].append({"cfactor": cfactor, "a": a, "raw": raw, "raw_to_float_depth": scale,
                                     "value": float(ref.CorrectDepth(F(cfactor), F(a), inv_depth))})
    for _ in range(16):
        p = [f32(rng.uniform(-1.5, 1.5)), f32(rng.uniform(-1.0, 1.0)), f32(rng.uniform(0.5, 4.0))]
        cam = [f32(rng.uniform(300, 600)), f32(rng.uniform(300, 600)), f32(rng.uniform(250, 400)), f32(rng.uniform(180, 300))]
        v = ref.Project(sympy.Matrix([F(c) for c in p]), *[F(c) for c in cam])
        out["project"].append({"point": p, "camera": cam, "value": [float(v[0]), float(v[1])]})
    for _ in range(16):
        cam = [f32(rng.uniform(300, 600)), f32(rng.uniform(300, 600)), f32(rng.uniform(250, 400)), f32(rng.uniform(180, 300))]
        x, y, depth = int(rng.integers(0, 640)), int(rng.integers(0, 480)), f32(rng.uniform(0.4, 6.0))
        # PixelCenterUnprojector's parameters (B/surfel_projection.h:61-71): 1/fx, 1/fy, -(cx - 0.5)/fx, -(cy - 0.5)/fy
        fx_inv, fy_inv = 1 / F(cam[0]), 1 / F(cam[1])
        cx_inv, cy_inv = -(F(cam[2]) - sympy.Rational(1, 2)) * fx_inv, -(F(cam[3]) - sympy.Rational(1, 2)) * fy_inv
        v = ref.Unproject(x, y, F(depth), fx_inv, fy_inv, cx_inv, cy_inv)
        out["unproject"].append({"camera": cam, "x": x, "y": y, "depth": depth, "value": [float(v[0]), float(v[1]), float(v[2])]})
    for _ in range(16):
        texels = [int(t) for t in rng.integers(0, 256, 4)]                     # top-left, top-right, bottom-left, bottom-right
        x, y = f32(rng.uniform(0.02, 0.98)), f32(rng.uniform(0.02, 0.98))
        v = ref.InterpolateBilinear(F(x), F(y), *[sympy.Rational(t, 255) for t in texels])
        out["bilinear"].append({"texels": texels, "x": x, "y": y, "value": float(v)})
    for _ in range(16):
        # the script implements a small-angle branch meant to be differentiated at zero (its real quaternion part is
        # 1 - theta^2 / 2 instead of the half-angle cosine 1 - theta^2 / 8): exact to 1e-8 only for |omega| of a few 1e-3
        omega = [f32(c) for c in rng.uniform(-0.002, 0.002, 3)]
        R = ref.SO3exp(sympy.Matrix([F(c) for c in omega]))
        out["so3_exp"].append({"omega": omega, "value": [float(R[i, j]) for i in range(3) for j in range(3)]})
    doc = {"source": "applications/badslam/scripts/jacobians_derivation.py (CorrectDepth, Project, Unproject, InterpolateBilinear, SO3exp)",
           "generator": "scripts/make_golden_functions.py", "precision_digits": PREC, "cases": out}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "functions.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", path, {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
