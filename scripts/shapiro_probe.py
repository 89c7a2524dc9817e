#!/usr/bin/env python3
"""Developer tool: launches the Shapiro filter of u, v, h (dynamics.py:610-626) at 721x1440 on every step, for a list of strip
heights; run under `rocprofv3 --kernel-trace --output-format csv` and read the kernel durations with scripts/trace_summary.py
(workgroup counts tell the strip heights apart).  python scripts/shapiro_probe.py [R ...]; QD_SHAPIRO_STREAM=0: per-pass kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    rs = [int(a) for a in sys.argv[1:]] or [16]
    grid, m, oc, forcing, mask, base_albedo, friction = bench.build_case(721, 1440, False)
    dev = m._dev
    dev.params.shapiro_every = 1
    dev.push_params()
    stars = forcing.star_table([i * 300.0 for i in range(8 * (len(rs) + 1))])
    dev.step_n(stars[:8], 300.0, with_ocean=False, with_physics=False, pass_albedo=True)
    for k, R in enumerate(rs):
        os.environ["QD_SHAPIRO_R"] = str(R)
        dev.lib.qd_tune_reload(dev.h)
        dev.step_n(stars[8 * (k + 1):8 * (k + 2)], 300.0, with_ocean=False, with_physics=False, pass_albedo=True)
    dev.sync()


if __name__ == "__main__":
    main()
