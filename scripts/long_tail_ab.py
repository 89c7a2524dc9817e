"""Developer check: 600 coupled steps at 181 x 360 with k_ocn_tail_fast's slim waves, with every wave on the general form (must be
bit-identical), and with the round-3 kernel (another strip cut = another order of the eta sum: after hundreds of steps the runs
have diverged by O(1) through the polar-row instability tests/test_oracle_polar_noise_cpu.py measures -- printed for the record).
   python scripts/long_tail_ab.py"""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.getcwd())
def run(env):
    code = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, bench
grid, m, oc, forcing, mask, base_albedo, friction = bench.build_case(181, 360, True)
dev = m._dev
dt = 300.0
n = 600
dev.step_n(forcing.star_table([i * dt for i in range(n)]), dt, with_ocean=True, with_physics=True, pass_albedo=True)
out = {k: np.array(dev.get(k)) for k in ("UO","VO","ETA","SST","TS","U","V","H","Q","CLOUD")}
np.savez(sys.argv[1], **out)
print("nsub", dev.last_ocean_nsub())
'''
    e = dict(os.environ); e.update(env)
    f = "/tmp/long_%s.npz" % ("_".join(f"{k}{v}" for k, v in env.items()) or "def")
    r = subprocess.run([sys.executable, "-c", code, f], env=e, capture_output=True, text=True)
    print(env, r.stdout.strip()[-60:], r.stderr.strip()[-200:])
    return np.load(f)
a = run({}); b = run({"QD_TAIL_GENERAL": "1"}); c = run({"QD_TAIL_V": "1"})
for k in a.files:
    same = np.array_equal(a[k], b[k], equal_nan=True)
    rel = float(np.nanmax(np.abs(a[k] - c[k])) / (np.nanmax(np.abs(c[k])) + 1e-300))
    print(k, "fast == general:", same, "| vs round-3 kernel rel", f"{rel:.2e}", "finite", bool(np.isfinite(a[k]).all()))
