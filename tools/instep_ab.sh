#!/bin/bash
# In-step A/B of COMPILE-TIME variants: each variant = one source recompiled with -D flags and linked into a private library under /tmp (the repo's library is
# never touched); bench.py then runs the overlapped step on it through XP_LIB_PATH.  Parameters tuned on stand-alone launches have repeatedly come out
# differently in the three-stream step, so this is the harness for them.
#   usage (GPU box, repo root): INSTEP_VARIANTS="name|file.hip|-DX=1 -DY=2;name2|..." [INSTEP_ARGS="--precision-class amp16f"] bash tools/instep_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/instep_ab.txt; mkdir -p $R/gpurun_out
T=/tmp/instep_ab; rm -rf $T; mkdir -p $T/obj
J='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print(d["value"], d["ms_per_step"])'
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I $R/include -I $R/xpoint_amd/csrc"
run() { for i in 1 2; do XP_LIB_PATH=$1 python3 $R/bench.py --steps ${INSTEP_STEPS:-100} --warmup 3 --no-cpu-baseline --no-h2d --no-other-backend $INSTEP_ARGS 2>/dev/null | python3 -c "$J"; done; }
echo "== baseline (repo library) $INSTEP_ARGS" | tee -a $OUT; run $R/xpoint_amd/libxpoint_hip.so | tee -a $OUT
IFS=';' read -ra VARS <<< "$INSTEP_VARIANTS"
for v in "${VARS[@]}"; do
  IFS='|' read -r name file defs <<< "$v"
  cp $R/xpoint_amd/csrc/_obj/*.o $T/obj/
  hipcc -x hip -c $R/xpoint_amd/csrc/$file -o $T/obj/$file.o $FLAGS $defs 2>/dev/null || { echo "build failed: $name" | tee -a $OUT; continue; }
  hipcc -shared -fPIC --offload-arch=gfx950 -o $T/lib_$name.so $T/obj/*.o
  echo "== $name ($file $defs) $INSTEP_ARGS" | tee -a $OUT; run $T/lib_$name.so | tee -a $OUT
done
echo "== baseline again" | tee -a $OUT; run $R/xpoint_amd/libxpoint_hip.so | tee -a $OUT
