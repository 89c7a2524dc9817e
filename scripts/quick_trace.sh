#!/bin/bash
# one rocprofv3 kernel-trace pass of a short bench run -> gpurun_out/<tag>/kernel_trace_summary.txt + step_timeline.txt
#   bash scripts/quick_trace.sh tag [bench.py args]
set -e -o pipefail
TAG=${1:-qt}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --no-cpu-baseline --no-ecology-leg --steps 96 --warmup 12 --timing-stride 100000 "$@" > $OUT/bench_under_trace.json 2> $OUT/trace.err
cd $ROOT
python3 scripts/trace_summary.py $OUT/trace > $OUT/kernel_trace_summary.txt
python3 scripts/step_timeline.py $OUT/trace k_dyn --seq > $OUT/step_timeline.txt
rm -rf $OUT/trace/t_kernel_trace.csv
