"""Test / bench infrastructure (like everything under oracle/): one full cost evaluation of a scene by the reference's own
functions (oracle/_ref), in a process of its own -- no HIP runtime, no torch in the address space -- for bench.py's
cpu_baseline.  usage: python -m oracle.ref_cost_worker <directory written by bench.py>  ->  one JSON line on stdout."""
import json
import os
import sys
import time

import numpy as np


def main():
    d = sys.argv[1]
    meta = json.load(open(os.path.join(d, "meta.json")))
    from oracle import binding as ob
    from oracle import ref_binding as rb
    W, H, K, N = meta["width"], meta["height"], meta["keyframes"], meta["surfels"]
    cam = np.asarray(meta["camera"], np.float32)
    orc = ob.OracleBA(N + 64, meta["raw_to_float_depth"], meta["baseline_fx"], meta["cell"], ob.make_camera(cam, W, H), ob.make_camera(cam, W, H))
    images = {name: np.load(os.path.join(d, name + ".npy"), mmap_mode="r") for name in ("depth", "normals", "radius", "color")}
    poses = np.load(os.path.join(d, "poses.npy"))
    for k in range(K):
        orc.add_preprocessed_keyframe(images["depth"][k], images["normals"][k], images["radius"][k], images["color"][k], poses[k])
    data = np.load(os.path.join(d, "surfels.npy"), mmap_mode="r")
    orc.surfel_data[:data.shape[0], :N] = data
    orc.surfels.surfels_size = orc.surfels.surfel_count = N
    t0 = time.time()
    cost, nres = rb.evaluate_cost(orc)
    dt = time.time() - t0
    print(json.dumps({"seconds": dt, "cost": cost, "nres": nres, "cores": int(ob.lib().orc_num_threads())}))


if __name__ == "__main__":
    main()
