#!/bin/bash
# Self-ring A/B of the two transports (one band of 8 on ONE GPU; developer rehearsal, never a reported number):
#   scripts/peer_selfring_ab.sh <outdir> [steps] [warmup]
# peer exchange with the three forms of QD_PEER_OVERLAP (0 no split, 1 push as a kernel of its own, 2 push carried by the interior launch), then RCCL
out=${1:-gpurun_out/selfring}; K=${2:-48}; W=${3:-12}
mkdir -p $out
for grid in "721 1440" "1441 2880"; do
  set -- $grid
  for ov in 2 1 0; do
    QD_PEER_EXCHANGE=1 QD_PEER_OVERLAP=$ov QD_BENCH_SELF_RING=8 timeout -k 10 300 python bench.py --nlat $1 --nlon $2 --steps $K --warmup $W --no-cpu-baseline --no-ecology-leg \
      > $out/selfring8_$1x$2_peer1_ov$ov.json 2> $out/selfring8_$1x$2_peer1_ov$ov.err || exit 1
  done
  QD_PEER_EXCHANGE=0 QD_BENCH_SELF_RING=8 timeout -k 10 300 python bench.py --nlat $1 --nlon $2 --steps $K --warmup $W --no-cpu-baseline --no-ecology-leg \
    > $out/selfring8_$1x$2_peer0.json 2> $out/selfring8_$1x$2_peer0.err || exit 1
done
