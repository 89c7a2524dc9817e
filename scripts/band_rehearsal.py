import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qingdai_amd as qa
from qingdai_amd.bands import BandGroup, required_halo
from qingdai_amd.device import Device
from qingdai_amd.topography import create_land_sea_mask, generate_base_properties
nlat,nlon,world,nsteps=(int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (721, 1440, 8, 12)))
grid=qa.SphericalGrid(nlat,nlon)
mask=create_land_sea_mask(grid); alb,fric=generate_base_properties(mask)
p=qa.QdParams(energy_w=1.0)
forcing=qa.ThermalForcing(qa.SphericalGrid(nlat,nlon), qa.OrbitalSystem())
stars=forcing.star_table([i*300.0 for i in range(nsteps)])
static={"LAND_MASK":mask,"FRICTION":fric,"BASE_ALBEDO":alb}
names=["U","V","H","TS","Q","CLOUD","UO","VO","ETA","SST"]
dev=Device(grid,p)
for k,v in static.items(): dev.upload_now(k,v)
dev.step_n(stars[:2],300.0,with_ocean=True,with_physics=True); dev.sync()
t0=time.perf_counter(); dev.step_n(stars[2:],300.0,with_ocean=True,with_physics=True); dev.sync(); t1=time.perf_counter()
ref={k:dev.get(k).copy() for k in names}; print("single ms/step",(t1-t0)/(nsteps-2)*1e3, "nsub", dev.last_ocean_nsub()); dev.close()
grp=BandGroup(grid,world,p); print("halo",grp.halo, grp.ranges)
for k,v in static.items(): grp.set(k,v)
grp.run(lambda d,r: d.step_n(stars[:2],300.0,with_ocean=True,with_physics=True))
e0=grp.exchanges()[0]; a0=grp.allreduces()[0]
t0=time.perf_counter(); grp.run(lambda d,r: (d.step_n(stars[2:],300.0,with_ocean=True,with_physics=True), d.sync())); t1=time.perf_counter()
print("bands ms/step (1 GPU, threads)",(t1-t0)/(nsteps-2)*1e3, "exchanges/step", (grp.exchanges()[0]-e0)/(nsteps-2),
      "all-reduces/step", (grp.allreduces()[0]-a0)/(nsteps-2))
for k in names:
    g=grp.get(k); s=max(float(np.max(np.abs(ref[k]))),1e-300)
    print(k, float(np.max(np.abs(g-ref[k])))/s)
