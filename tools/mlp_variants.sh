#!/bin/bash
# Stage-removal builds of the fused block-tail kernels (csrc/mlp_fused.hip, XP_MLP_DBG): one extra library per variant, built HERE (no GPU needed) so that a
# single gpurun call can time them all.   tools/mlp_variants.sh build | run
set -u
cd "$(dirname "$0")/.."
SET=${MLP_DBG_SET:-1 2 8 16 9 25}
if [ "${1:-build}" = build ]; then
  FLAGS=$(python3 -c "from xpoint_amd import build; print(' '.join(build.FLAGS))")
  for d in $SET; do
    ( hipcc -x hip -c xpoint_amd/csrc/mlp_fused.hip -o /tmp/mlp_dbg$d.o $FLAGS -DXP_MLP_DBG=$d 2>/dev/null || { echo "build failed ($d)"; exit 1; }
      OBJS=$(ls xpoint_amd/csrc/_obj/*.o | grep -v mlp_fused.hip.o)
      hipcc -shared -fPIC --offload-arch=gfx950 -o xpoint_amd/libxpoint_hip_mlpdbg$d.so $OBJS /tmp/mlp_dbg$d.o ) &
  done
  wait; ls -la xpoint_amd/libxpoint_hip_mlpdbg*.so
else
  mkdir -p gpurun_out
  { echo "== XP_MLP_DBG=0"; MLP_H2=1 MLP_FUSED_ONLY=1 MLP_ONLY=0,1 python tools/mlp_bench.py 2>&1 | grep "^M"
    for d in $SET; do echo "== XP_MLP_DBG=$d   (1 no GELU, 2 no LDS-DMA after the prologue, 4 no barriers, 8 no split of the hidden values, 16 no MFMA)"
      XP_LIB_PATH=$PWD/xpoint_amd/libxpoint_hip_mlpdbg$d.so MLP_H2=1 MLP_FUSED_ONLY=1 MLP_ONLY=0,1 python tools/mlp_bench.py 2>&1 | grep "^M"; done; } > gpurun_out/mlp_h2_stage_removal.txt
  cat gpurun_out/mlp_h2_stage_removal.txt
fi
