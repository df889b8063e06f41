#!/bin/bash
# Stage removal of nms_localmax_kernel on the GPU box (private copy of the library built with -DXP_NMS_DBG=<mask>; results are wrong on purpose).
#   usage: bash tools/nms_dbg.sh "0 1 2 4 8"
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=/tmp/nmsdbg; rm -rf $T; mkdir -p $T; cp -r $R/xpoint_amd $R/include $R/tools $T/; cd $T
for m in ${1:-0 1 2 4 8}; do
  hipcc -x hip -c xpoint_amd/csrc/postproc.hip -o xpoint_amd/csrc/_obj/postproc.hip.o --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I include -I xpoint_amd/csrc -DXP_NMS_DBG=$m 2>/dev/null
  hipcc -shared -fPIC --offload-arch=gfx950 -o xpoint_amd/libxpoint_hip.so xpoint_amd/csrc/_obj/*.o
  echo "== XP_NMS_DBG=$m (1 no candidate tests, 2 no out store, 4 no list building, 8 no loads, 16 no window walk)"
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/nst
  PYTHONPATH=$T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nst -- python3 $T/tools/nms_bench.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/nst/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "nms_" in r["Name"]: print("   ", r["Name"][:58], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
  cd $T
done
