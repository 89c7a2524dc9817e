#!/bin/bash
# Would a one-launch ocean sub-step pay on a BAND-sized grid?  Whole-globe handle of a band's shape (205 x 2880 = 1/8 of 1441 x 2880 + halos):
# two launches (QD_OCN_FUSED=0) against k_ocn_fused over strip heights.   scripts/fused_band_shape.sh <outdir>
out=${1:-gpurun_out/fband}; mkdir -p $out
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --nlat 205 --nlon 2880 --steps 48 --warmup 12 --no-cpu-baseline --no-ecology-leg > $out/$tag.json 2>$out/$tag.err || exit 1
  python - <<PY
import json
j=json.loads(open("$out/$tag.json").read().strip().splitlines()[-1]); print("$tag", round(j["ms_per_step"],4), "n_sub", j["config"].get("ocean_n_sub"))
PY
}
run two QD_OCN_FUSED=0
for r in 20 26 34 42 52 68; do run fused_r$r QD_OCN_FUSED=1 QD_FUSED_R=$r; done
