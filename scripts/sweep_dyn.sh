#!/bin/bash
# strip-height sweep of the momentum kernels in the 240-step bench:  bash scripts/sweep_dyn.sh [nlat nlon "extra bench args"]
set -e
NLAT=${1:-721}; NLON=${2:-1440}; EXTRA=${3:-}
run() { lab=$1; shift
  env "$@" timeout -k 10 250 python bench.py --no-cpu-baseline --no-ecology-leg --nlat $NLAT --nlon $NLON $EXTRA > gpurun_out/sd_$lab.json 2> gpurun_out/sd_$lab.err
  python - <<PY
import json; d=json.loads(open("gpurun_out/sd_$lab.json").read().strip().splitlines()[-1]); r=d["roofline"]; o=d.get("roofline_ocean_substep",{}); print("$lab", round(d["ms_per_step"],4), "dyn", round(r["avg_kernel_ms"]*1e3,2), round(r["frac"],3), "ocn", round(o.get("avg_kernel_ms",0)*1e3,2))
PY
}
run def A=1
for r in 16 20 28 32 40; do run d$r QD_STREAM_R_DYN=$r; done
for v in 0 3 9; do run vb$v QD_STREAM_VB=$v; done
