#!/bin/bash
# where does a small-K GEMM spend its time: K loop vs epilogue (wrong results on purpose)
for flags in "" "-DXP_EPI_DBG=1" "-DXP_X3_DBG=10" "-DXP_X3_DBG=10 -DXP_EPI_DBG=1" "-DXP_X3_DBG=62 -DXP_EPI_DBG=1"; do
  touch xpoint_amd/csrc/gemm_x3.hip
  XP_EXTRA_HIPCC_FLAGS="$flags" python -m xpoint_amd.build > /dev/null 2>&1 || echo build failed
  echo "== flags: $flags"
  GB_X3=1 GB_ONLY=${GB_ONLY:-3,2,0,8} python tools/gemm_bench.py 2>&1 | grep "^M"
done
touch xpoint_amd/csrc/gemm_x3.hip xpoint_amd/csrc/gemm.hip
python -m xpoint_amd.build > /dev/null 2>&1
