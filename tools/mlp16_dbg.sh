#!/bin/bash
# Stage removal for the fast class's fused MLP (results wrong on purpose).  Variants are built into a PRIVATE copy under /tmp: the repo's library is never touched.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/mlp16_stage_removal.txt; mkdir -p $R/gpurun_out; : > $OUT
T=/tmp/mlp16dbg; rm -rf $T; mkdir -p $T
cp -r $R/xpoint_amd $R/include $R/tools $T/
cd $T
for d in ${MLP16_DBG_LIST:-0 1}; do
  hipcc -x hip -c xpoint_amd/csrc/mlp_f16.hip -o xpoint_amd/csrc/_obj/mlp_f16.hip.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I include -I xpoint_amd/csrc -DXP_MLP16_DBG=$d 2>/dev/null || { echo "build failed for $d" >> $OUT; continue; }
  hipcc -shared -fPIC --offload-arch=gfx950 -o xpoint_amd/libxpoint_hip.so xpoint_amd/csrc/_obj/*.o
  echo "== XP_MLP16_DBG=$d   (1 GELU -> identity)" >> $OUT
  PYTHONPATH=$T python3 tools/mlp16_bench.py 2>&1 | grep "^M" >> $OUT
done
cat $OUT
