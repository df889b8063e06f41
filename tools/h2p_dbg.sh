#!/bin/bash
# Stage-removal timing of the ping-pong split-fp16 GEMM (gemm_h2p.hip) on the GPU box: rebuilds the one translation unit with -DXP_H2P_DBG=<mask> into a private copy
# of the library (results are WRONG by construction; timing only).   usage: bash tools/h2p_dbg.sh "0 1 2 4 6 7" [shape indices of tools/gemm_bench.py]
MASKS=${1:-"0 1 2 4 6 7"}
ONLY=${2:-"12,15,10,11"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=/tmp/h2pdbg; rm -rf $T; mkdir -p $T
cp -r $R/xpoint_amd $R/include $R/tools $T/
cd $T
for m in $MASKS; do
  hipcc -x hip -c xpoint_amd/csrc/gemm_h2p.hip -o xpoint_amd/csrc/_obj/gemm_h2p.hip.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I include -I xpoint_amd/csrc -DXP_H2P_DBG=$m $XP_H2P_EXTRA 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o xpoint_amd/libxpoint_hip.so xpoint_amd/csrc/_obj/*.o
  echo "== XP_H2P_DBG=$m   (1 no MFMA, 2 no staging, 4 no fragment reads, 8 no split VALU, 16 no LDS stores, 32 no global loads)"
  GB_H2=1 GB_ONLY=$ONLY PYTHONPATH=$T python3 tools/gemm_bench.py 2>&1 | grep -E "^M|stamps"
done
