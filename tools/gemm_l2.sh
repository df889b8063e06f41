#!/bin/bash
# Which resource does the deep-stage GEMM saturate?  L2-side request counters of ONE shape (default M 19200 x N 384 x K 1536 + residual: the ping-pong split-fp16
# kernel of the f32 class and the fp16-storage kernel of the fast class), next to the analytic L2 -> LDS tile traffic (tiles x slabs x bytes per slab).
#   usage (GPU box, repo root): bash tools/gemm_l2.sh  ->  gpurun_out/gemm_l2.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/gemm_l2; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export GB_SHAPES="${GB_SHAPES:-19200,384,1536,0,1}"
for mode in H2 F16; do
  for c in "TCC_REQ_sum TCC_READ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "TCC_EA0_RDREQ_sum"; do
    d=$OUT/${mode}_$(echo $c | tr ' ' '_')
    env GB_${mode}=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/gemm_bench.py > $d.log 2>&1 || echo "$mode $c failed: $(tail -1 $d.log)"
  done
done
python3 - "$OUT" "$R/gpurun_out/pmc_gemm_l2.json" > $R/gpurun_out/gemm_l2.txt <<'PY'
import collections, csv, glob, json, os, sys
L2_HIT_TBS, MALL_TBS = 17.8, 8.6          # MI355X_MICROARCH.md: reads served by the XCDs' L2s 16.8-18.8 TB/s (midpoint), by the Infinity Cache 8.6 TB/s
js = {"note": "rocprofv3 --pmc TCC_READ_sum / TCC_HIT_sum / TCC_MISS_sum / GRBM_GUI_ACTIVE / FETCH_SIZE on tools/gemm_bench.py, one shape; l2_read_tb_s = TCC_READ x 128 B / kernel "
              "time; ceiling = hit-rate-weighted blend of the guide's L2 (17.8) and Infinity-Cache (8.6 TB/s) read rates", "kernels": {}}
M, N, K = (int(v) for v in os.environ["GB_SHAPES"].split(";")[0].split(",")[:3])
print(f"# L2-side counters of one GEMM launch, M {M} x N {N} x K {K} (+ residual); per-launch averages over the micro-benchmark's 23 launches")
for mode, kern, desc in (("H2", "gemm_h2", "f32 class: split-fp16 (A f32 from HBM, B two fp16 planes)"), ("F16", "gemm_f16", "fast class: fp16 storage")):
    vals = collections.defaultdict(lambda: [0.0, 0])
    name = None
    for path in glob.glob(os.path.join(sys.argv[1], mode + "_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if kern in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0][:60]
                v = vals[r["Counter_Name"]]; v[0] += float(r["Counter_Value"]); v[1] += 1
    avg = {k: v[0] / v[1] for k, v in vals.items() if v[1]}
    tiles = -(-M // 128) * -(-N // 128)
    if mode == "H2": slab_bytes, nslab = (128 * 32 * 4) + (128 * 32 * 4), -(-K // 32)         # A: f32, 32-wide slab; B: two fp16 planes
    else: slab_bytes, nslab = (128 + 128) * 64 * 2, -(-K // 64)
    tile_mb = tiles * nslab * slab_bytes / 1e6
    cyc = avg.get("GRBM_GUI_ACTIVE", float("nan")) / 8.0
    us = cyc / 2400.0
    print(f"\n{desc}\n  kernel {name}")
    for k in sorted(avg): print(f"  {k:24s} {avg[k]:16.0f}")
    print(f"  kernel cycles {cyc:.0f} (= {us:.1f} us at 2.4 GHz)")
    print(f"  analytic L2 -> LDS tile traffic: {tiles} tiles x {nslab} slabs x {slab_bytes // 1024} KB = {tile_mb:.0f} MB per launch = {tile_mb / us:.2f} TB/s"
          f"  (MI355X_MICROARCH.md: 16.8-18.8 TB/s for reads served by the XCDs' L2s, 8.6 TB/s from the Infinity Cache)")
    if "TCC_READ_sum" in avg and "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
        hit = avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
        rd = avg["TCC_READ_sum"] * 128 / 1e6 / us
        ceil = 1.0 / (hit / L2_HIT_TBS + (1.0 - hit) / MALL_TBS)
        print(f"  L2 reads {rd:.2f} TB/s at hit rate {hit:.3f}: ceiling of that mix {ceil:.1f} TB/s -> {rd / ceil:.2f} of it  <- the saturated resource (matrix pipe: see tools/f16_pmc.sh / mfma_util.sh)")
        js["kernels"][kern] = {"shape": [M, N, K], "kernel_us": us, "l2_read_tb_s": rd, "l2_hit_rate": hit, "l2_ceiling_tb_s": ceil, "frac_of_l2_ceiling": rd / ceil,
                               "beyond_l2_mb": avg.get("FETCH_SIZE", float("nan")) * 2 * 1024 / 1e6, "tile_traffic_mb_analytic": tile_mb}
    if "TCC_READ_sum" in avg: print(f"  TCC_READ x 128 B = {avg['TCC_READ_sum'] * 128 / 1e6:.0f} MB (x 64 B = {avg['TCC_READ_sum'] * 64 / 1e6:.0f} MB)")
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg: print(f"  L2 hit rate {avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']):.3f}")
    if "FETCH_SIZE" in avg: print(f"  beyond L2 (FETCH_SIZE x 2 x 1024): {avg['FETCH_SIZE'] * 2 * 1024 / 1e6:.0f} MB")
json.dump(js, open(sys.argv[2], "w"), indent=1)
PY
find $OUT -name "*.csv" -size +2000k -delete
cat $R/gpurun_out/gemm_l2.txt
