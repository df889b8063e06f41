#!/bin/bash
# Build tools/ring_bench (XP_RING_DBG = 0) and its stage-removal variants; run here (CPU box: build only) or on the GPU box (run with `run`).
#   tools/ring_dbg.sh build      -> tools/ring_bench, tools/ring_bench_dbg{1,2,4,5,6}
#   tools/ring_dbg.sh run [args] -> runs all of them, output to gpurun_out/ring_*.txt
set -u
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I xpoint_amd/csrc -I include"
if [ "${1:-build}" = build ]; then
    hipcc $FLAGS tools/ring_bench.hip -o tools/ring_bench || { echo "build failed"; exit 1; }
    for d in ${RING_DBGS:-1 2 4 5 6}; do
        hipcc $FLAGS -DXP_RING_DBG=$d tools/ring_bench.hip -o tools/ring_bench_dbg$d || { echo "build failed (dbg $d)"; exit 1; } &
    done
    wait
    ls -la tools/ring_bench*
else
    shift
    mkdir -p gpurun_out
    tools/ring_bench "$@" > gpurun_out/ring_main.txt 2>&1
    for d in ${RING_DBGS:-1 2 4 5 6}; do
        [ -x tools/ring_bench_dbg$d ] && tools/ring_bench_dbg$d ${1:-0} ${2:-5} 0 > gpurun_out/ring_dbg$d.txt 2>&1
    done
    tail -n +1 gpurun_out/ring_main.txt
fi
