#!/bin/bash
# A/B of one environment switch on the default bench:   scripts/ab_env.sh <outdir> <VAR> <value A> <value B> [reps] [bench args]
out=$1; var=$2; a=$3; b=$4; reps=${5:-2}; shift 5 || shift $#
mkdir -p $out
for rep in $(seq $reps); do
  for v in $a $b; do
    env $var=$v timeout -k 10 300 python bench.py --steps 96 --warmup 24 --no-cpu-baseline --no-ecology-leg "$@" > $out/${var}_${v}_rep$rep.json 2> $out/${var}_${v}_rep$rep.err || exit 1
    echo "$var=$v rep $rep: $(grep -o '"ms_per_step": [0-9.]*' $out/${var}_${v}_rep$rep.json)"
  done
done
