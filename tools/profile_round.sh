#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh r01'
# Writes CSVs under gpurun_out/prof_<tag>_*; tools/summarise_profiles.py turns them into profiles/<tag>_*.
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace -- $BENCH > $R/gpurun_out/prof_${TAG}_trace.log 2>&1 || exit 1
SHORT="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- $SHORT > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- $SHORT > $R/gpurun_out/prof_${TAG}_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_sq -- $SHORT > $R/gpurun_out/prof_${TAG}_sq.log 2>&1 || exit 1
tail -1 $R/gpurun_out/prof_${TAG}_trace.log | cut -c1-300
echo "profiles collected for $TAG"
