#!/bin/bash
# A/B on one box: the bench's timed region with and without a live RCCL communicator (world 1), and HW queue count
run() { "$@" python bench.py --no-cpu-baseline --no-other-backend --no-h2d 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'])"; }
for i in 1 2; do
  echo "norccl            $(run env XP_BENCH_NO_RCCL=1)"
  echo "rccl              $(run env)"
  echo "rccl hwq8         $(run env GPU_MAX_HW_QUEUES=8)"
  echo "norccl hwq8       $(run env XP_BENCH_NO_RCCL=1 GPU_MAX_HW_QUEUES=8)"
done
