#!/bin/bash
# timing experiments for the split-bf16 GEMM K loop: rebuild with one stage removed at a time (results are wrong on purpose)
for d in 0 1 2 4 8 3 12; do
  touch xpoint_amd/csrc/gemm_x3.hip
  XP_EXTRA_HIPCC_FLAGS="-DXP_X3_DBG=$d" python -m xpoint_amd.build > /dev/null 2>&1 || echo build failed
  echo "== XP_X3_DBG=$d"
  GB_X3=1 GB_ONLY=${GB_ONLY:-8,12,14,2} python tools/gemm_bench.py 2>&1 | grep "^M"
done
touch xpoint_amd/csrc/gemm_x3.hip
python -m xpoint_amd.build > /dev/null 2>&1
