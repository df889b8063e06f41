#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) of the kernels matching $1 in a short single-stream bench run: MB per launch
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ktraffic; rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-backend --no-overlap --no-h2d > /dev/null 2>&1
done
python3 $R/tools/pmc_summary.py $(find $OUT/FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/WRITE_SIZE -name "*counter_collection.csv" | head -1) $OUT/t.json | grep -i "${1:-sample}"
find $OUT -name "*.csv" -size +2000k -delete
