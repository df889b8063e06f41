#!/bin/bash
# kernel trace of the default (overlapped) bench and of the single-stream one; GPU idle time and concurrency over the timed part
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_pipe; rm -rf $OUT; mkdir -p $OUT
for v in overlap single; do
  fl=""; [ $v = single ] && fl="--no-overlap"
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$v -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-other-backend --no-h2d $fl > /dev/null 2>&1
  echo "== $v"; python3 $R/tools/trace_gaps.py $(find $OUT/$v -name "*kernel_trace.csv" | head -1) 0.25
done
find $OUT -name "*.csv" -size +3000k -delete
