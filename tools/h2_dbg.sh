#!/bin/bash
# Stage removal for the split-fp16 tile engine's K loop: one stage compiled out at a time (results are wrong on purpose), two shapes timed.
# The debug variants are built into a PRIVATE copy under /tmp (like tools/h2p_dbg.sh): the repo's own libxpoint_hip.so is never touched, so an interrupted
# run cannot leave a wrong-results library in the tree.   usage (GPU box, repo root): bash tools/h2_dbg.sh  ->  gpurun_out/h2_stage_removal.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/h2_stage_removal.txt; mkdir -p $R/gpurun_out; : > $OUT
T=/tmp/h2dbg; rm -rf $T; mkdir -p $T
cp -r $R/xpoint_amd $R/include $R/tools $T/
cd $T
for d in ${H2_DBG_LIST:-0 1 2 4 8 16 32 36 6 14 30 62}; do
  hipcc -x hip -c xpoint_amd/csrc/gemm_h2.hip -o xpoint_amd/csrc/_obj/gemm_h2.hip.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I include -I xpoint_amd/csrc -DXP_H2_DBG=$d 2>/dev/null || { echo "build failed for $d" >> $OUT; continue; }
  hipcc -shared -fPIC --offload-arch=gfx950 -o xpoint_amd/libxpoint_hip.so xpoint_amd/csrc/_obj/*.o
  echo "== XP_H2_DBG=$d   (1 no split VALU, 2 no global loads, 4 no MFMA, 8 no LDS stores, 16 no barrier, 32 no fragment reads)" >> $OUT
  XP_H2P=0 GB_H2=1 GB_ONLY=${GB_ONLY:-12,15,10} PYTHONPATH=$T python3 tools/gemm_bench.py 2>&1 | grep "^M" >> $OUT
done
cat $OUT
