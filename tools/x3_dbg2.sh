#!/bin/bash
# single-workgroup-per-CU K-loop timeline of the split-bf16 GEMM with stages removed (wrong results on purpose)
for flags in "" "-DXP_X3_DBG=2" "-DXP_X3_DBG=8" "-DXP_X3_DBG=10" "-DXP_X3_DBG=4" "-DXP_X3_DBG=16" "-DXP_X3_DBG=18" "-DXP_X3_DBG=26" "-DXP_X3_DBG=32" "-DXP_X3_DBG=42" "-DXP_X3_DBG=58"; do
  touch xpoint_amd/csrc/gemm_x3.hip
  XP_EXTRA_HIPCC_FLAGS="$flags" python -m xpoint_amd.build > /dev/null 2>&1 || echo build failed
  echo "== flags: $flags"
  GB_X3=1 GB_ONLY=${GB_ONLY:-20,12} python tools/gemm_bench.py 2>&1 | grep "^M"
done
touch xpoint_amd/csrc/gemm_x3.hip
python -m xpoint_amd.build > /dev/null 2>&1
