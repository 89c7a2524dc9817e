#!/bin/bash
# Developer tool: 721x1440 bench with k_ocn_fused (QD_OCN_FUSED=1) over strip heights; scripts/fused_sweep.sh <outdir> R...
out=$1; shift
mkdir -p $out
for r in "$@"; do
  QD_OCN_FUSED=1 QD_FUSED_R=$r timeout -k 10 300 python bench.py --steps 48 --warmup 12 --no-cpu-baseline --no-ecology-leg --profile-kernel ocean_step > $out/fused_r$r.json 2>$out/fused_r$r.err || exit 1
  python - <<PY
import json
j=json.loads(open("$out/fused_r$r.json").read().strip().splitlines()[-1]); print("R=$r", round(j["ms_per_step"],4), j["config"]["ocean_n_sub"], "kernel us", round(j["roofline"]["avg_kernel_ms"]*1e3,2))
PY
done
