"""TEST TOOL (not collected by pytest): how long the device loop and the oracle stay together on the coupled configuration.
   python tests/long_run_vs_oracle.py 91 180 240   (GPU box)"""
import sys, time
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))
for _p in (ROOT, ROOT + '/tests', ROOT + '/oracle'): sys.path.insert(0, _p)
import numpy as np
import qd_oracle as qo
from qd_oracle.driver import DriverOracle
from qingdai_amd.driver import Simulation
import qingdai_amd as qa
from util import relerr
nlat, nlon, nsteps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sim = Simulation(nlat, nlon, params=qa.QdParams(), use_ocean=True, quiet=True, ecology=False)
g = qo.Grid(nlat, nlon); P = qo.defaults()
m = qo.AtmosOracle(g, sim.friction, sim.land_mask, P, C_s_map=np.where(sim.land_mask == 1, 3e6, P.Cs_ocean).astype(float))
oc = qo.OceanOracle(g, sim.land_mask, P, init_Ts=np.full((nlat, nlon), 288.0))
d = DriverOracle(g, m, oc, qo.Forcing(g), sim.land_mask, sim.base_albedo, P)
done = 0
for chunk in (1, 3, 8, 12, 24, 48, 96, 144, 240):
    if chunk > nsteps: break
    n = chunk - done
    sim.run_steps(n)
    t0 = time.time()
    for i in range(done, chunk): d.step(i * 300.0, 300)
    done = chunk
    pairs = {"u": (sim.gcm.u, m.u), "v": (sim.gcm.v, m.v), "h": (sim.gcm.h, m.h), "T_s": (sim.gcm.T_s, m.T_s), "q": (sim.gcm.q, m.q),
             "cloud": (sim.gcm.cloud_cover, m.cloud_cover), "uo": (sim.ocean.uo, oc.uo), "eta": (sim.ocean.eta, oc.eta), "SST": (sim.ocean.Ts, oc.Ts),
             "precip": (sim.dev.get("PRECIP"), d.precip)}
    print(chunk, {k: f"{relerr(a, b):.1e}" for k, (a, b) in pairs.items()}, f"oracle {time.time()-t0:.1f}s", flush=True)
