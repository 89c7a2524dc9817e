#!/usr/bin/env python3
"""Developer tool: the coupled 721x1440 loop with a list of strip heights of k_ocn_tail_stream (QD_TAIL_R is read at create; qd_tune_reload re-reads it);
run under `rocprofv3 --kernel-trace --output-format csv` and read the durations with scripts/trace_summary.py (the workgroup counts
tell the strip heights apart).  python scripts/tail_probe.py [R ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    rs = [int(a) for a in sys.argv[1:]] or [8]
    grid, m, oc, forcing, mask, base_albedo, friction = bench.build_case(721, 1440, True)
    dev = m._dev
    stars = forcing.star_table([i * 300.0 for i in range(24 + 6 * len(rs))])
    dev.step_n(stars[:24], 300.0, with_ocean=True, with_physics=True, pass_albedo=True)
    for k, R in enumerate(rs):
        os.environ["QD_TAIL_R"] = str(R)
        dev.lib.qd_tune_reload(dev.h)
        dev.step_n(stars[24 + 6 * k:30 + 6 * k], 300.0, with_ocean=True, with_physics=True, pass_albedo=True)
    dev.sync()


if __name__ == "__main__":
    main()
