#!/bin/bash
# Developer tool: strip heights of the two ocean kernels on ONE band of eight (self-ring over the peer exchange), 1441x2880 by default.
#   scripts/band_sweep.sh <outdir> [nlat nlon]
out=${1:-gpurun_out/band_sweep}; nlat=${2:-1441}; nlon=${3:-2880}
mkdir -p $out
run() { tag=$1; shift
  env QD_PEER_EXCHANGE=1 QD_BENCH_SELF_RING=8 "$@" timeout -k 10 300 python bench.py --nlat $nlat --nlon $nlon --steps 48 --warmup 12 --no-cpu-baseline --no-ecology-leg > $out/$tag.json 2>$out/$tag.err || { echo "$tag FAILED"; return; }
  python - <<PY
import json
j=json.loads(open("$out/$tag.json").read().strip().splitlines()[-1]); print("$tag", round(j["ms_per_step"],4), j["config"]["ocean_n_sub"])
PY
}
run auto
for r in 20 28 40 60; do run ocn$r QD_STREAM_R_OCN=$r; done
for r in 10 14 20 28; do run tail$r QD_TAIL_R=$r; done
for r in 24 36 60; do run dyn$r QD_STREAM_R_DYN=$r; done
