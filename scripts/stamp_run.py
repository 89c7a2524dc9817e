import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
grid, m, oc, forcing, mask, base_albedo, friction = bench.build_case(721, 1440, True)
dev = m._dev
dt = 300.0
dev.step_n(forcing.star_table([i * dt for i in range(6)]), dt, with_ocean=True, with_physics=True, pass_albedo=True)
dev.sync()
