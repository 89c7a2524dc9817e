#!/usr/bin/env python3
"""Drop-in for the reference's `python -m scripts.run_simulation` (scripts/run_simulation.py:1161):
same QD_* environment surface for the per-timestep path, restart NetCDF files and signal handling;
the loop itself runs on the MI355X (qingdai_amd.driver)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from qingdai_amd.driver import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
