#!/usr/bin/env python3
"""Occupancy over time of the two sweeps, from an experiment build with -DBAHIP_TILE_TIMELINE (make -C badslam_amd/csrc variant
NAME=timeline EXTRA=-DBAHIP_TILE_TIMELINE; BADSLAM_LIB_DIR=.../lib_variants/timeline BAHIP_TIMELINE_DIR=<dir> python bench.py
--no-extras --no-cpu-baseline): every tile of the last launch wrote its start and end time (100 MHz clock).  Prints how long the
launch lasted, how much of it ran with fewer than 90 % / 50 % of the peak number of tiles in flight (the tail), the spread of
the tile durations and what a perfectly balanced launch would have taken.
usage: tile_timeline.py <dir>"""
import os
import sys

import numpy as np


def analyse(name, path):
    words = 4 if name == "pose" else 2                   # pose records also carry the tile and its candidate counts
    a = np.fromfile(path, np.uint64).reshape(-1, words)
    a = a[(a[:, 0] > 0) & (a[:, 1] >= a[:, 0])]
    if len(a):
        a = a[a[:, 0] > a[:, 1].max() - 200000]           # the last launch only (positions an earlier launch used may be left over)
    if len(a) == 0:
        print(f"{name}: no records")
        return
    t0 = a[:, 0].min()
    start = (a[:, 0] - t0).astype(np.float64) * 1e-2      # microseconds
    end = (a[:, 1] - t0).astype(np.float64) * 1e-2
    dur = end - start
    total = end.max()
    # tiles in flight over time
    ev = np.concatenate([np.stack([start, np.ones_like(start)], 1), np.stack([end, -np.ones_like(end)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    inflight = np.cumsum(ev[:, 1])
    dt = np.diff(ev[:, 0], append=ev[-1, 0])
    peak = inflight.max()
    busy = float((inflight * dt).sum())                   # tile-microseconds
    below90 = float(dt[inflight < 0.9 * peak].sum())
    below50 = float(dt[inflight < 0.5 * peak].sum())
    print(f"{name}: {len(a)} tiles, launch {total:.0f} us, peak {int(peak)} tiles in flight, mean in flight {busy / total:.0f}")
    print(f"  time below 90 % of the peak: {below90:.0f} us ({100 * below90 / total:.1f} %), below 50 %: {below50:.0f} us ({100 * below50 / total:.1f} %)")
    print(f"  tile duration us: mean {dur.mean():.1f}  median {np.median(dur):.1f}  p90 {np.percentile(dur, 90):.1f}  p99 {np.percentile(dur, 99):.1f}  max {dur.max():.1f}")
    print(f"  perfectly balanced at the peak occupancy: {busy / peak:.0f} us ({100 * (1 - busy / peak / total):.1f} % of the launch is imbalance / ramp)")
    # where do the long tiles start?
    order = np.argsort(start)
    q = len(order) // 10
    for label, idx in (("first 10 %", order[:q]), ("middle", order[4 * q:6 * q]), ("last 10 %", order[-q:])):
        print(f"  tiles started in the {label}: mean duration {dur[idx].mean():.1f} us, max {dur[idx].max():.1f} us")
    if words == 4:
        visited, assoc = (a[:, 3] >> np.uint64(32)).astype(np.float64), (a[:, 3] & np.uint64(0xffffffff)).astype(np.float64)
        ok = visited > 0
        print(f"  candidates per tile: visited mean {visited.mean():.1f} max {visited.max():.0f}, with an association mean {assoc.mean():.1f}; "
              f"us per visited candidate: median {np.median(dur[ok] / visited[ok]):.2f}, p99 {np.percentile(dur[ok] / visited[ok], 99):.2f}; "
              f"correlation of duration with visited {np.corrcoef(dur[ok], visited[ok])[0, 1]:.3f}, with associated {np.corrcoef(dur[ok], assoc[ok])[0, 1]:.3f}")
    last = np.argsort(end)[-5:]
    print("  the five tiles that ended last: started at", " ".join(f"{start[i]:.0f}" for i in last), "us, lasted", " ".join(f"{dur[i]:.0f}" for i in last), "us")


if __name__ == "__main__":
    d = sys.argv[1]
    for name in ("geometry", "pose"):
        p = os.path.join(d, f"{name}_timeline.bin")
        if os.path.exists(p):
            analyse(name, p)
