#!/bin/bash
# Developer tool: k_ocn_fused at 1441x2880 over strip heights (and the two launches for comparison); scripts/fused_sweep_big.sh <outdir> R...
out=$1; shift
mkdir -p $out
one() { tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --nlat 1441 --nlon 2880 --steps 24 --warmup 6 --no-cpu-baseline --no-ecology-leg --profile-kernel ocean_step > $out/$tag.json 2>$out/$tag.err || { echo "$tag FAILED"; return; }
  python - <<PY
import json
j=json.loads(open("$out/$tag.json").read().strip().splitlines()[-1]); print("$tag", round(j["ms_per_step"],4), j["config"]["ocean_n_sub"], "ocean_step kernel us", round(j["roofline"]["avg_kernel_ms"]*1e3,2))
PY
}
one two_launches QD_OCN_FUSED=0
for r in "$@"; do one fused_r$r QD_OCN_FUSED=1 QD_FUSED_R=$r; done
