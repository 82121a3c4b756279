#!/bin/bash
# PMC passes (SQ issue / wait / LDS counters, then FETCH_SIZE and WRITE_SIZE each in a pass of its own) over ANY python
# workload of this repository, aggregated per FrAD kernel:
#   gpurun --timeout 600 -- 'bash tools/pmc_any.sh p1 tools/p1_probe.py'
# -> gpurun_out/pmc_<tag>.json {kernel: {counter: mean per launch, ..., hbm_bytes_per_launch_corrected}}
# Only --pmc + --kernel-trace are combined (the pool refuses --pmc with the runtime / sys traces).
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SCRIPT=$R/$1; shift
run() {  # leg name, counters...
    local leg=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_${leg} -- python3 $SCRIPT $ARGS > $R/gpurun_out/pmc_${TAG}_${leg}.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${TAG}_${leg}.log; exit 1; }
}
ARGS="$*"
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
run f FETCH_SIZE
run w WRITE_SIZE
python3 $R/tools/pmc_aggregate.py $R ${TAG}
