#!/bin/bash
# Developer tool: tune_stream.py over the A/B builds in tools/variants (tools/build_variant.sh).  bash scripts/sweep_variants.sh "<tags>" "<R list>"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAGS=${1:-base}
RS=${2:-0 24 36 48}
mkdir -p $ROOT/gpurun_out
for t in $TAGS; do
  if [ "$t" = base ]; then unset QD_LIB_PATH; else export QD_LIB_PATH=$ROOT/tools/variants/libqingdai_hip_$t.so; fi
  echo "== variant $t"
  python3 $ROOT/scripts/tune_stream.py $RS || exit 1
done
