#!/bin/bash
# per-kernel times of the matcher micro-benchmark: bash tools/match_prof.sh <tag> [n] [cap] [pairs]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-m}; shift
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -- python3 $R/tools/match_bench.py "$@" 2>/dev/null | grep "us per call"
python3 $R/tools/kstats.py $(find $R/gpurun_out/$TAG -name "*kernel_stats.csv" | head -1) 30 | grep -i match
