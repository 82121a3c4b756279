#!/bin/bash
# rocprofv3 --kernel-trace --stats of the profile-1 device chain (K7, Golomb encode / compaction, Golomb decode, K8,
# overlap-add) on the cfg-5 geometry, 60 s and 10 min clips:  gpurun --timeout 600 -- 'bash tools/profile_p1.sh r02'
# -> gpurun_out/prof_<tag>_p1/ ; condensed by hand into profiles/<tag>_p1_kernel_stats.csv (frad kernels only).
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_p1 -- python3 $R/tools/p1_chain.py > $R/gpurun_out/prof_${TAG}_p1.log 2>&1 || exit 1
F=$(ls -t $R/gpurun_out/prof_${TAG}_p1/*/*_kernel_stats.csv | head -1)
head -1 $F > $R/gpurun_out/${TAG}_p1_kernel_stats.csv
grep "frad::" $F >> $R/gpurun_out/${TAG}_p1_kernel_stats.csv
cat $R/gpurun_out/${TAG}_p1_kernel_stats.csv | cut -c1-160
