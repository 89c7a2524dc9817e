#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: run on the GPU box from the repo root,   bash scripts/profile_round.sh r03 [bench.py args]
# e.g.   bash scripts/profile_round.sh r03_1441x2880 --nlat 1441 --nlon 2880 --steps 48 --warmup 12
# One pass per counter group (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc never with --stats).
set -e -o pipefail
TAG=${1:-rXX}
shift || true
EXTRA="$@"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-ecology-leg $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B > $OUT/bench_under_trace.json 2> $OUT/trace.err
echo "trace pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o t -- $B --steps 12 --warmup 4 > /dev/null 2> $OUT/fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o t -- $B --steps 12 --warmup 4 > /dev/null 2> $OUT/write.err
echo "WRITE_SIZE pass done"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -o t -- $B --steps 12 --warmup 4 > /dev/null 2> $OUT/sq.err
echo "SQ pass done"
cd $ROOT
python3 scripts/trace_summary.py $OUT/trace > $OUT/kernel_trace_summary.txt
python3 scripts/step_timeline.py $OUT/trace k_dyn > $OUT/step_timeline.txt
python3 scripts/pmc_summary.py $OUT/fetch $OUT/fetch.json > /dev/null
python3 scripts/pmc_summary.py $OUT/write $OUT/write.json > /dev/null
python3 scripts/pmc_summary.py $OUT/sq $OUT/sq.json > /dev/null
cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null || true
rm -rf $OUT/trace/t_kernel_trace.csv $OUT/fetch/*.csv $OUT/write/*.csv $OUT/sq/*.csv      # the raw traces are tens of MB
python3 bench.py $EXTRA > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
