#!/bin/bash
# Stage removal for the split-fp16 tile engine's K loop: rebuild with one stage compiled out at a time (results are wrong on purpose), time two shapes.
#   usage (GPU box, repo root): bash tools/h2_dbg.sh  ->  gpurun_out/h2_stage_removal.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/h2_stage_removal.txt; : > $OUT
for d in ${H2_DBG_LIST:-0 1 2 4 8 16 32 36 6 14 30 62}; do
  touch $R/xpoint_amd/csrc/gemm_h2.hip
  XP_EXTRA_HIPCC_FLAGS="-DXP_H2_DBG=$d" python3 -m xpoint_amd.build > /dev/null 2>&1 || echo "build failed for $d" >> $OUT
  echo "== XP_H2_DBG=$d   (1 no split VALU, 2 no global loads, 4 no MFMA, 8 no LDS stores, 16 no barrier, 32 no fragment reads)" >> $OUT
  GB_H2=1 GB_ONLY=${GB_ONLY:-12,15,10} python3 $R/tools/gemm_bench.py 2>&1 | grep "^M" >> $OUT
done
touch $R/xpoint_amd/csrc/gemm_h2.hip
python3 -m xpoint_amd.build > /dev/null 2>&1
cat $OUT
