#!/bin/bash
# per-sweep durations of the NMS fixed-point kernel under different local-iteration schedules (rocprofv3 kernel trace)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sc in "8" "2,2,8" "2,3,8" "3,3,8" "2,2,4,8" "1,2,4,8" "2,2,2,8"; do
  rm -rf $R/gpurun_out/kt
  XP_NMS_SCHED=$sc rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-backend --no-overlap > $R/gpurun_out/kt.log 2>&1
  python3 - "$sc" <<PY
import csv,glob,os,sys
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/kt/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "nms_sweep" in r["Kernel_Name"]]
ok = "not converged" not in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/kt.log").read()
print("sched", sys.argv[1].ljust(10), "sweeps (us):", [round(x,1) for x in d[-6:]], "sum", round(sum(d[-6:]),1), "converged" if ok else "NOT CONVERGED")
PY
done
rm -rf $R/gpurun_out/kt $R/gpurun_out/kt.log
