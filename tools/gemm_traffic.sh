#!/bin/bash
# HBM traffic of the split-fp16 GEMM per model shape: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/gemm_bench.py, against the algorithmic bytes
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/gemm_traffic; rm -rf $OUT; mkdir -p $OUT
export GB_H2=1 GB_SHAPES="${GB_SHAPES:-19200,384,384,0,0;19200,1536,384,1,0;19200,384,1536,0,1;4800,3072,768,1,0;4800,768,3072,0,1}"
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/tools/gemm_bench.py > $OUT/$c.log 2>&1 || echo "$c failed"
done
python3 - <<PY
import csv, glob, collections
shapes = [tuple(int(v) for v in t.split(',')) for t in "$GB_SHAPES".split(';')]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if "gemm_h2_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c] if f else []
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    vals[c] = [float(r["Counter_Value"]) for r in rows]
per = len(vals["FETCH_SIZE"]) // len(shapes)          # launches per shape (warm-up + timed)
for i, (M, N, K, act, res) in enumerate(shapes):
    sl = slice(i * per, (i + 1) * per)
    avg = lambda c: sum(vals[c][sl]) / max(len(vals[c][sl]), 1) if vals[c] else float("nan")
    alg_r = 4.0 * (M * K + N * K + (M * N if res else 0)); alg_w = 4.0 * M * N
    print(f"M{M} N{N} K{K} res{res}: fetch {avg('FETCH_SIZE') * 1024 / 1e6:7.1f} MB x2 = {2 * avg('FETCH_SIZE') * 1024 / 1e6:7.1f} (algorithmic reads {alg_r / 1e6:6.1f} MB)  write {avg('WRITE_SIZE') * 1024 / 1e6:7.1f} MB (algorithmic {alg_w / 1e6:6.1f})  L2 hit {avg('TCC_HIT_sum'):.3g} miss {avg('TCC_MISS_sum'):.3g}")
PY
find $OUT -name "*.csv" -size +2000k -delete
