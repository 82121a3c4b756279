#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r03'
# 1. --kernel-trace --stats of the default bench.py run                     -> gpurun_out/prof_<tag>_trace/
# 2. PMC passes (SQ issue / LDS / wait counters, FETCH_SIZE, WRITE_SIZE each in a pass of its own; only --pmc + --kernel-trace
#    are ever combined) of a short bench.py run                             -> gpurun_out/pmc_<tag>_bench.json
# 3. the same passes of tools/pmc_workload.py: K7, K8, Golomb, cfg 4, the mixed-radix tails, overlap-add -> gpurun_out/pmc_<tag>_other.json
# tools/summarise_profiles.py <tag> condenses them into profiles/<tag>_kernel_stats.csv and profiles/<tag>_counters.json.
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_trace.log 2>&1 || exit 1
grep "^{\"metric\"" $R/gpurun_out/prof_${TAG}_trace.log | tail -1 > $R/gpurun_out/prof_${TAG}_bench_line.json   # (rocprofv3 prints its own closing lines after the program's)
echo "trace done"
cd $R
bash tools/pmc_any.sh ${TAG}_bench bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_bench.log 2>&1 || { tail -3 $R/gpurun_out/pmc_${TAG}_bench.log; exit 1; }
echo "bench counters done"
bash tools/pmc_any.sh ${TAG}_other tools/pmc_workload.py > $R/gpurun_out/pmc_${TAG}_other.log 2>&1 || { tail -3 $R/gpurun_out/pmc_${TAG}_other.log; exit 1; }
echo "profiles collected for $TAG"
