#!/usr/bin/env python3
"""Developer tool: HIP-event time of the two fused kernels (k_dyn_hyper / k_ocn_hyper groups) at 721x1440 for a list of
strip heights of the row-streaming kernels (QD_STREAM_R is read at create; qd_tune_reload re-reads it).  python scripts/tune_stream.py [R ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402


def main():
    rs = [int(a) for a in sys.argv[1:]] or [12, 16, 20, 24, 30, 36, 48]
    nlat, nlon = int(os.environ.get("QD_TUNE_NLAT", "721")), int(os.environ.get("QD_TUNE_NLON", "1440"))
    grid, m, oc, forcing, mask, base_albedo, friction = bench.build_case(nlat, nlon, True)
    dev = m._dev
    if os.environ.get("QD_TUNE_COPYONLY") == "1":           # k4 <= 0: every field passes through (loads + momentum + stores only)
        dev.params.sigma4 = 0.0; dev.params.sigma4_ocean = 0.0
        dev.push_params()
    dt = 300.0
    stars = forcing.star_table([i * dt for i in range(400)])
    dev.step_n(stars[:24], dt, with_ocean=True, with_physics=True, pass_albedo=True)
    dev.sync()
    k = 24
    for R in rs:
        if R > 0:
            os.environ["QD_STREAM_R"] = str(R)
        else:
            os.environ.pop("QD_STREAM_R", None)                  # 0: the library's own choice
        dev.lib.qd_tune_reload(dev.h)
        dev.timing(select="k_dyn_hyper,k_ocn_hyper")
        t0 = time.perf_counter()
        dev.step_n(stars[k:k + 12], dt, with_ocean=True, with_physics=True, pass_albedo=True)
        dev.sync()
        el = (time.perf_counter() - t0) / 12 * 1e3
        k += 12
        d_ms, d_n = dev.timing_get("k_dyn_hyper")
        o_ms, o_n = dev.timing_get("k_ocn_hyper")
        dev.timing(on=False)
        cells = nlat * nlon
        print(f"R={R:3d}  k_dyn {d_ms * 1e3:7.2f} us ({88.0 * cells / d_ms / 1e6 / 8000:.3f} of 8 TB/s, n={d_n})   "
              f"k_ocn {o_ms * 1e3:7.2f} us ({65.0 * cells / o_ms / 1e6 / 8000:.3f}, n={o_n})   step {el:.3f} ms  n_sub {dev.last_ocean_nsub()}",
              flush=True)


if __name__ == "__main__":
    main()
