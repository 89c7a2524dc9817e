#!/usr/bin/env python3
"""Developer tool: HIP-event time of the ocean tail launch (timer group ocean_tail) and of the whole step at 721x1440 for a list of
strip heights of the streaming tail kernel (QD_TAIL_R is read at create; qd_tune_reload re-reads it).  python scripts/tune_tail.py [R ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    rs = [int(a) for a in sys.argv[1:]] or [8, 12, 16, 24, 32]
    nlat, nlon = int(os.environ.get("QD_TUNE_NLAT", "721")), int(os.environ.get("QD_TUNE_NLON", "1440"))
    grid, m, oc, forcing, mask, base_albedo, friction = bench.build_case(nlat, nlon, True)
    dev = m._dev
    dt = 300.0
    stars = forcing.star_table([i * dt for i in range(600)])
    dev.step_n(stars[:24], dt, with_ocean=True, with_physics=True, pass_albedo=True)
    dev.sync()
    k = 24
    for R in rs:
        os.environ["QD_TAIL_R"] = str(R)
        dev.lib.qd_tune_reload(dev.h)
        dev.timing(select="ocean_tail,k_ocn_hyper,ocean_step")
        t0 = time.perf_counter()
        dev.step_n(stars[k:k + 12], dt, with_ocean=True, with_physics=True, pass_albedo=True)
        dev.sync()
        el = (time.perf_counter() - t0) / 12 * 1e3
        k += 12
        t_ms, t_n = dev.timing_get("ocean_tail")
        o_ms, o_n = dev.timing_get("k_ocn_hyper")
        s_ms, s_n = dev.timing_get("ocean_step")
        dev.timing(on=False)
        print(f"R={R:3d}  tail {t_ms * 1e3:7.2f} us (n={t_n})   k_ocn {o_ms * 1e3:7.2f} us (n={o_n})   ocean_step {s_ms * 1e3:7.2f} us (n={s_n})   "
              f"step {el:.3f} ms  n_sub {dev.last_ocean_nsub()}", flush=True)


if __name__ == "__main__":
    main()
