#!/bin/bash
# A/B of the ocean tail kernel in the 240-step bench (same box):  bash scripts/ab_tail.sh "label ENV=.. ENV=.." ...
set -e
for spec in "$@"; do
  set -- $spec; lab=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-ecology-leg > gpurun_out/ab_$lab.json 2> gpurun_out/ab_$lab.err
  python - <<PY
import json; d=json.loads(open("gpurun_out/ab_$lab.json").read().strip().splitlines()[-1]); print("$lab", round(d["ms_per_step"],4), round(d["value"],3))
PY
done
