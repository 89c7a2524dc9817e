#!/usr/bin/env python3
"""How far do the three in-step medians move from one step to the next?  (decides the width of a speculative bracket)
   python scripts/median_drift.py [nlat nlon steps]   -> per step and site: median, relative move, candidates, positives"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

nlat = int(sys.argv[1]) if len(sys.argv) > 1 else 721
nlon = int(sys.argv[2]) if len(sys.argv) > 2 else 1440
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
grid, m, oc, forcing, mask, base_albedo, friction = bench.build_case(nlat, nlon, True)
dev = m._dev
dt = 300.0
buf = (ctypes.c_double * 64)()
last = [None] * 4
worst = [0.0] * 4
for i in range(steps):
    stars = forcing.star_table([i * dt])
    dev.step_n(stars, dt, with_ocean=True, with_physics=True, pass_albedo=True)
    dev.lib.qd_median_state(dev.h, buf)
    line = [f"step {i:3d}"]
    for s in (1, 2, 3):
        med, cand, npos, hits, miss = buf[16 * s], buf[16 * s + 6], buf[16 * s + 7], buf[16 * s + 4], buf[16 * s + 5]
        rel = abs(med / last[s] - 1.0) if last[s] else float("nan")
        if i >= 3 and rel == rel: worst[s] = max(worst[s], rel)
        last[s] = med
        line.append(f"| s{s} {med:.6e} move {rel:8.2e} cand {cand:6.0f}/{npos:7.0f} h{hits:.0f} m{miss:.0f}")
    print(" ".join(line), flush=True)
print("worst relative move after step 3:", worst[1:])
