#!/bin/bash
# Patch-shape sweep of the fp16 depthwise convolution (PH x PW pixels per thread).  Variants are built into a PRIVATE copy under /tmp.
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/dwconv16_patch_sweep.txt; mkdir -p $R/gpurun_out; : > $OUT
T=/tmp/dw16dbg; rm -rf $T; mkdir -p $T
cp -r $R/xpoint_amd $R/include $R/tools $T/
cd $T
for v in ${DW16_LIST:-"4 4" "2 4" "4 2" "2 2" "2 8" "1 4" "1 8"}; do
  set -- $v
  hipcc -x hip -c xpoint_amd/csrc/elementwise_f16.hip -o xpoint_amd/csrc/_obj/elementwise_f16.hip.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I include -I xpoint_amd/csrc -DXP_DWH_PH=$1 -DXP_DWH_PW=$2 2>/dev/null || { echo "build failed for $v" >> $OUT; continue; }
  hipcc -shared -fPIC --offload-arch=gfx950 -o xpoint_amd/libxpoint_hip.so xpoint_amd/csrc/_obj/*.o
  echo "== PH=$1 PW=$2" >> $OUT
  PYTHONPATH=$T python3 tools/dwconv16_bench.py 2>&1 | grep "^H\|^sum" >> $OUT
done
cat $OUT
