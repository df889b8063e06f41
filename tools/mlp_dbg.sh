#!/bin/bash
# timing experiments for the fused MLP: rebuild with one stage removed at a time (results are wrong on purpose)
for d in ${MLP_DBG_SET:-0 1 9 2 4 6 16 15}; do
  touch xpoint_amd/csrc/mlp_fused.hip
  XP_EXTRA_HIPCC_FLAGS="-DXP_MLP_DBG=$d" python -m xpoint_amd.build > /dev/null 2>&1 || echo build failed
  echo "== XP_MLP_DBG=$d"
  python tools/mlp_bench.py 2>&1 | grep "^M" | head -1
done
touch xpoint_amd/csrc/mlp_fused.hip
python -m xpoint_amd.build > /dev/null 2>&1
