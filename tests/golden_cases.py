"""tests/golden/jacobians.json turned into calls of the five Jacobian functions (oracle: orc_jac_*, HIP: bahip_debug_jacobian).
The golden side works in the reference's variables (global normal, global_T_frame, ...); the functions take what the kernels
have at hand (normal and points in the keyframe frame, image gradients).  The conversion is plain linear algebra in binary64.

Each case: (name, kind, inputs, expected, picked_outputs) with `kind` / `inputs` in the layout of bahip_debug_jacobian."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jacobians.json")


def bilinear_gradient(texels, px, py):
    """Derivative of the bilinear lookup wrt the pixel position (what B/cost_function.cuh:200-211 samples)."""
    tl, tr, bl, br = texels
    fx, fy = px - np.floor(px), py - np.floor(py)
    return (1 - fy) * (tr - tl) + fy * (br - bl), (1 - fx) * (bl - tl) + fx * (br - tr)


def load():
    with open(GOLDEN) as f:
        return json.load(f)


def jacobian_cases():
    g = load()["cases"]
    out = []
    for c in g["depth_pose"]:
        G = np.array(c["global_T_frame"]).reshape(3, 4)
        nl = G[:, :3].T @ np.array(c["surfel_normal"])            # surfel normal in the keyframe frame
        out.append(("depth_pose", 0, list(nl) + list(c["local_point"]) + [1.0], c["jacobian"], [0, 1, 2, 3, 4, 5]))
    for c in g["depth_intrinsics"]:
        G = np.array(c["global_T_frame"]).reshape(3, 4)
        nl = G[:, :3].T @ np.array(c["surfel_normal"])
        out.append(("depth_intrinsics", 3, [c["x"], c["y"], c["depth"], 1.0, nl[0], nl[1], 0.0, 0.0, 1.0, 1.0, 1.0], c["jacobian"], [0, 1, 2, 3]))
    for c in g["depth_correction"]:
        G = np.array(c["global_T_frame"]).reshape(3, 4)
        nl = G[:, :3].T @ np.array(c["surfel_normal"])
        fx_inv, fy_inv, cx_inv, cy_inv = c["intrinsics"]
        nx, ny = fx_inv * c["x"] + cx_inv, fy_inv * c["y"] + cy_inv
        exp_inv_depth = float(np.exp(-c["a"] * c["raw_inv_depth"]))
        corrected = c["cfactor"] * exp_inv_depth + c["raw_inv_depth"]
        # golden order: cfactor, a; function rows: [4] = a, [5] = cfactor
        out.append(("depth_correction", 3, [c["x"], c["y"], 1.0 / corrected, 1.0, nl[0], nl[1], float(np.dot([nx, ny, 1.0], nl)), c["cfactor"],
                                            c["raw_inv_depth"], exp_inv_depth, corrected], c["jacobian"], [5, 4]))
    for c in g["descriptor_pose"]:
        ls = np.array(c["local_surfel_pos"])
        fx, fy, cx, cy = c["camera"]
        gx, gy = bilinear_gradient(c["texels"], fx * ls[0] / ls[2] + cx, fy * ls[1] / ls[2] + cy)
        out.append(("descriptor_pose", 1, list(ls) + [gx * fx, gy * fy], c["jacobian"], [0, 1, 2, 3, 4, 5]))   # gradient times fx, fy
    for c in g["descriptor_surfel"]:
        Fm = np.array(c["frame_T_global"]).reshape(3, 4)
        lp = Fm[:, :3] @ np.array(c["surfel_pos"]) + Fm[:, 3]
        rn = Fm[:, :3] @ np.array(c["surfel_normal"])
        fx, fy, cx, cy = c["camera"]
        gx, gy = bilinear_gradient(c["texels"], fx * lp[0] / lp[2] + cx, fy * lp[1] / lp[2] + cy)
        out.append(("descriptor_surfel", 2, list(rn) + list(lp) + [gx, gy, fx, fy], c["jacobian"], [0]))
    for c in g["descriptor_color_intrinsics"]:
        ls = np.array(c["local_surfel_pos"])
        fx, fy, cx, cy = c["camera"]
        gx, gy = bilinear_gradient(c["texels"], fx * ls[0] / ls[2] + cx, fy * ls[1] / ls[2] + cy)
        out.append(("descriptor_color_intrinsics", 4, [gx, gy, ls[0] / ls[2], ls[1] / ls[2]], c["jacobian"], [0, 1, 2, 3]))
    return out


def close(got, want, rel=2e-5):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert np.abs(got - want).max() <= rel * max(1.0, np.abs(want).max()), (got, want)
