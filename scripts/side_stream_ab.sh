#!/bin/bash
# A/B of QD_SIDE_STREAM (the next step's precipitation block beside the ocean sub-steps instead of in front of them):
#   scripts/side_stream_ab.sh <outdir> [steps] [warmup]
out=${1:-gpurun_out/side}; K=${2:-96}; W=${3:-24}
mkdir -p $out
for rep in 1 2; do
  for s in 0 1; do
    QD_SIDE_STREAM=$s timeout -k 10 300 python bench.py --steps $K --warmup $W --no-cpu-baseline --no-ecology-leg \
      > $out/side${s}_rep$rep.json 2> $out/side${s}_rep$rep.err || exit 1
  done
done
QD_SIDE_STREAM=1 timeout -k 10 300 python bench.py --nlat 1441 --nlon 2880 --steps 24 --warmup 8 --no-cpu-baseline --no-ecology-leg > $out/side1_big.json 2> $out/side1_big.err || exit 1
QD_SIDE_STREAM=0 timeout -k 10 300 python bench.py --nlat 1441 --nlon 2880 --steps 24 --warmup 8 --no-cpu-baseline --no-ecology-leg > $out/side0_big.json 2> $out/side0_big.err || exit 1
grep -h -o '"ms_per_step": [0-9.]*' $out/side*.json
