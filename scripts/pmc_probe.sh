#!/bin/bash
# Counter passes for the memory pipeline of the streaming kernels (one rocprofv3 --pmc run per group, never with --stats):
#   bash scripts/pmc_probe.sh tag   -> gpurun_out/<tag>/<group>.json (mean per dispatch and kernel)
set -e -o pipefail
TAG=${1:-pmc}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-ecology-leg --steps 8 --warmup 4 --timing-stride 100000"
run() {
  name=$1; shift
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o t -- $B > /dev/null 2> $OUT/$name.err || { echo "$name FAILED"; tail -3 $OUT/$name.err; return 0; }
  python3 $ROOT/scripts/pmc_summary.py $OUT/$name $OUT/$name.json stream > /dev/null
  rm -rf $OUT/$name
  echo "$name done"
}
if [ -n "$QD_PMC_ONLY" ]; then
  run $QD_PMC_ONLY GRBM_GUI_ACTIVE $QD_PMC_COUNTERS
  exit 0
fi
# at most two counters of a block per pass (more: "Request exceeds the capabilities of the hardware", and the aborted tool hangs)
run ta1 GRBM_GUI_ACTIVE TA_TA_BUSY TA_BUFFER_TOTAL_CYCLES
run ta2 GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
run tcp1 GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY
run tcp2 GRBM_GUI_ACTIVE TCP_TCC_READ_REQ TCP_TCP_TA_DATA_STALL_CYCLES
run tlb GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT
run tcc1 GRBM_GUI_ACTIVE TCC_BUSY TCC_TAG_STALL
run tcc2 GRBM_GUI_ACTIVE TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL
run sqv GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
