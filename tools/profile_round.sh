#!/bin/bash
# Evidence run for profiles/: bench line (with cpu_baseline), rocprofv3 kernel stats and PMC HBM traffic of the same command.
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r1_03
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/hip_event_breakdown.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-backend --no-overlap --no-h2d > $OUT/bench_under_rocprof.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-backend --no-overlap --no-h2d > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-backend --no-overlap --no-h2d > /dev/null 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 $R/tools/pmc_summary.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json > $OUT/pmc_hbm_traffic.txt
rm -rf $OUT/stats $OUT/fetch $OUT/write
cat $OUT/bench.json; head -12 $OUT/kernel_stats.csv | cut -c1-200; head -20 $OUT/pmc_hbm_traffic.txt
