#!/bin/bash
# Round-6 experiment builds of the fused block tail (csrc/mlp_fused.hip): software-pipelined chunk loop for the split-fp16 instances (XP_MLP_H2_PIPE), MFMA
# accumulators in AccVGPRs (XP_MLP_AGPR), waves per SIMD the kernel is compiled for (XP_MLP_WPE).  One extra library per variant, built HERE (no GPU needed).
#   tools/mlp_pipe_variants.sh build | run
set -u
cd "$(dirname "$0")/.."
declare -A V=( [pipe_v2]="-DXP_MLP_H2_PIPE=1" [pipe_a2]="-DXP_MLP_H2_PIPE=1 -DXP_MLP_AGPR=1" [pipe_a1]="-DXP_MLP_H2_PIPE=1 -DXP_MLP_AGPR=1 -DXP_MLP_WPE=1" [ser_a1]="-DXP_MLP_AGPR=1 -DXP_MLP_WPE=1" )
if [ "${1:-build}" = build ]; then
  FLAGS=$(python3 -c "from xpoint_amd import build; print(' '.join(build.FLAGS))")
  for k in "${!V[@]}"; do
    ( hipcc -x hip -c xpoint_amd/csrc/mlp_fused.hip -o /tmp/mlp_$k.o $FLAGS ${V[$k]} -Rpass-analysis=kernel-resource-usage 2>/tmp/mlp_$k.log || { echo "build failed ($k)"; tail -5 /tmp/mlp_$k.log; exit 1; }
      OBJS=$(ls xpoint_amd/csrc/_obj/*.o | grep -v mlp_fused.hip.o)
      hipcc -shared -fPIC --offload-arch=gfx950 -o xpoint_amd/libxpoint_hip_$k.so $OBJS /tmp/mlp_$k.o ) &
  done
  wait; ls -la xpoint_amd/libxpoint_hip_*.so
else
  mkdir -p gpurun_out
  for rep in 1 2; do
    echo "== default"; MLP_H2=1 MLP_FUSED_ONLY=1 MLP_ONLY=${MLP_ONLY:-0,1} MLP_CRC=1 python tools/mlp_bench.py 2>&1 | grep -E "^M|CRC|Error|error" | tail -8
    for k in "${!V[@]}"; do echo "== $k  (${V[$k]})"
      XP_LIB_PATH=$PWD/xpoint_amd/libxpoint_hip_$k.so MLP_H2=1 MLP_FUSED_ONLY=1 MLP_ONLY=${MLP_ONLY:-0,1} MLP_CRC=1 python tools/mlp_bench.py 2>&1 | grep -E "^M|CRC|Error|error" | tail -8; done
  done
fi
