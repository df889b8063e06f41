#!/bin/bash
# A/B builds of csrc/mlp_fused.hip with extra -D flags: one extra library per variant, built HERE (no GPU needed).
#   tools/mlp_ab.sh build <tag> "<flags>"      -> xpoint_amd/libxpoint_hip_mlpab_<tag>.so
#   tools/mlp_ab.sh run <tag> [<tag> ...]       -> alternating timings + CRCs of tools/mlp_bench.py (MLP_H2=1), the default library first
set -u
cd "$(dirname "$0")/.."
if [ "${1:-}" = build ]; then
  tag=$2; shift 2
  FLAGS=$(python3 -c "from xpoint_amd import build; print(' '.join(build.FLAGS))")
  hipcc -x hip -c xpoint_amd/csrc/mlp_fused.hip -o /tmp/mlp_ab_$tag.o $FLAGS "$@" 2>/tmp/mlp_ab_$tag.log || { echo "build failed ($tag)"; tail -5 /tmp/mlp_ab_$tag.log; exit 1; }
  OBJS=$(ls xpoint_amd/csrc/_obj/*.o | grep -v mlp_fused.hip.o)
  hipcc -shared -fPIC --offload-arch=gfx950 -o xpoint_amd/libxpoint_hip_mlpab_$tag.so $OBJS /tmp/mlp_ab_$tag.o && ls -la xpoint_amd/libxpoint_hip_mlpab_$tag.so
else
  shift
  for rep in 1 2 3; do
    echo "== default"; MLP_CRC=1 MLP_H2=1 MLP_FUSED_ONLY=1 MLP_ONLY=${MLP_ONLY:-0,1} python tools/mlp_bench.py 2>&1 | grep "^M\|^CRC"
    for tag in "$@"; do echo "== $tag"; XP_LIB_PATH=$PWD/xpoint_amd/libxpoint_hip_mlpab_$tag.so MLP_CRC=1 MLP_H2=1 MLP_FUSED_ONLY=1 MLP_ONLY=${MLP_ONLY:-0,1} python tools/mlp_bench.py 2>&1 | grep "^M\|^CRC"; done
  done
fi
