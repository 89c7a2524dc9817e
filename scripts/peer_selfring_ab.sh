#!/bin/bash
# Self-ring A/B of the two transports (one band of 8 on ONE GPU; developer rehearsal, never a reported number):
#   scripts/peer_selfring_ab.sh <outdir> [steps] [warmup]
out=${1:-gpurun_out/selfring}; K=${2:-48}; W=${3:-12}
mkdir -p $out
for grid in "721 1440" "1441 2880"; do
  set -- $grid
  for t in 1 0; do
    QD_PEER_EXCHANGE=$t QD_BENCH_SELF_RING=8 timeout -k 10 300 python bench.py --nlat $1 --nlon $2 --steps $K --warmup $W --no-cpu-baseline --no-ecology-leg \
      > $out/selfring8_$1x$2_peer$t.json 2> $out/selfring8_$1x$2_peer$t.err || exit 1
  done
done
