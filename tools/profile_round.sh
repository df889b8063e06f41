#!/bin/bash
# Evidence run for profiles/: bench lines (c2 with cpu_baseline, c4, c5), rocprofv3 kernel stats, PMC HBM traffic and MFMA utilisation of the
# same commands.   usage (on the GPU box, from the repo root): bash tools/profile_round.sh r2_03
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/hip_event_breakdown.txt
python3 $R/bench.py --config c4 --steps 30 --no-cpu-baseline > $OUT/c4_bench.json 2> $OUT/c4_hip_event_breakdown.txt
python3 $R/bench.py --config c5 --steps 30 --no-cpu-baseline > $OUT/c5_bench.json 2> $OUT/c5_hip_event_breakdown.txt
COMMON="--no-cpu-baseline --no-other-backend --no-overlap --no-h2d"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 $COMMON > $OUT/bench_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats4 -- python3 $R/bench.py --config c4 --steps 5 --warmup 2 $COMMON > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5 -- python3 $R/bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 $COMMON > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 $COMMON > /dev/null 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats4 -name "*kernel_stats.csv" | head -1) $OUT/c4_kernel_stats.csv
cp $(find $OUT/stats5 -name "*kernel_stats.csv" | head -1) $OUT/c5_kernel_stats.csv
python3 $R/tools/pmc_summary.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json c2 > $OUT/pmc_hbm_traffic.txt
rm -rf $OUT/stats $OUT/stats4 $OUT/stats5 $OUT/fetch $OUT/write
# round 6: the C4 / C5 lines quote their OWN counters (per-configuration summaries, each stamped with the kernel-source hash: bench.py pmc_files / pmc_fields)
for CF in c4 c5; do
  EXTRA="--no-cpu-baseline --no-other-backend --no-h2d"; [ $CF = c4 ] && EXTRA="$EXTRA --no-overlap"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$CF -- python3 $R/bench.py --config $CF --steps 2 --warmup 1 $EXTRA > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$CF -- python3 $R/bench.py --config $CF --steps 2 --warmup 1 $EXTRA > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $(find $OUT/fetch_$CF -name "*counter_collection.csv" | head -1) $(find $OUT/write_$CF -name "*counter_collection.csv" | head -1) $OUT/${CF}_pmc_traffic.json $CF > $OUT/${CF}_pmc_hbm_traffic.txt
  cp $OUT/${CF}_pmc_traffic.json $R/profiles/pmc_traffic_$CF.json
  OV="--no-overlap"; [ $CF = c5 ] && OV=" "
  XP_MFMA_UTIL_OVERLAP="$OV" XP_MFMA_UTIL_ARGS="--config $CF" XP_MFMA_UTIL_TAG=$CF bash $R/tools/mfma_util.sh > /dev/null 2>&1
  cp $R/gpurun_out/mfma_utilisation_$CF.txt $OUT/${CF}_mfma_utilisation.txt; cp $R/gpurun_out/pmc_mfma_$CF.json $OUT/${CF}_pmc_mfma.json; cp $R/gpurun_out/pmc_mfma_$CF.json $R/profiles/pmc_mfma_$CF.json
  rm -rf $OUT/fetch_$CF $OUT/write_$CF
done
bash $R/tools/mfma_util.sh > /dev/null 2>&1
cp $R/gpurun_out/mfma_utilisation.txt $OUT/
cp $R/gpurun_out/pmc_mfma.json $OUT/; cp $R/gpurun_out/pmc_mfma.json $R/profiles/pmc_mfma.json
python3 $R/tools/match_bench.py 4060 8192 8 > $OUT/match_microbench.txt 2>&1
# the fast mixed-precision class (gemm_mode amp16f: half storage): its own bench line, kernel stats and MFMA-busy counters — never the headline
python3 $R/bench.py --precision-class amp16f --no-cpu-baseline --no-h2d > $OUT/amp16f_bench.json 2> $OUT/amp16f_hip_event_breakdown.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats16 -- python3 $R/bench.py --precision-class amp16f --steps 5 --warmup 2 $COMMON > /dev/null 2>&1
cp $(find $OUT/stats16 -name "*kernel_stats.csv" | head -1) $OUT/amp16f_kernel_stats.csv; rm -rf $OUT/stats16
XP_MFMA_UTIL_ARGS="--precision-class amp16f" XP_MFMA_UTIL_TAG=amp16f bash $R/tools/mfma_util.sh > /dev/null 2>&1
cp $R/gpurun_out/mfma_utilisation_amp16f.txt $OUT/amp16f_mfma_utilisation.txt; cp $R/gpurun_out/pmc_mfma_amp16f.json $OUT/amp16f_pmc_mfma.json
cp $R/gpurun_out/pmc_mfma_amp16f.json $R/profiles/pmc_mfma_amp16f.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch16 -- python3 $R/bench.py --precision-class amp16f --steps 2 --warmup 1 $COMMON > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write16 -- python3 $R/bench.py --precision-class amp16f --steps 2 --warmup 1 $COMMON > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(find $OUT/fetch16 -name "*counter_collection.csv" | head -1) $(find $OUT/write16 -name "*counter_collection.csv" | head -1) $OUT/amp16f_pmc_traffic.json amp16f > $OUT/amp16f_pmc_hbm_traffic.txt
rm -rf $OUT/fetch16 $OUT/write16
cp $OUT/amp16f_pmc_traffic.json $R/profiles/pmc_traffic_amp16f.json
# the class's line again with its own counter files in place (roofline.traffic, frac_mfma_busy_pmc)
python3 $R/bench.py --precision-class amp16f --no-cpu-baseline --no-h2d > $OUT/amp16f_bench.json 2> $OUT/amp16f_hip_event_breakdown.txt
GB_F16=1 python3 $R/tools/gemm_bench.py > $OUT/gemm_f16_microbench.txt 2>&1
GB_H2=1 python3 $R/tools/gemm_bench.py > $OUT/gemm_h2_microbench.txt 2>&1
GB_X3=1 python3 $R/tools/gemm_bench.py > $OUT/gemm_x3_microbench.txt 2>&1
python3 $R/tools/scan_bench.py > $OUT/selective_scan_microbench.txt 2>&1
python3 $R/tools/ss2d_bench.py 0 1 2>&1 | grep -v amdgpu > $OUT/ss2d_core_microbench.txt
python3 $R/tools/enc_only.py 2>&1 | grep stream > $OUT/encoder_only_streams.txt
# round 5: the ring dense engine's stand-alone table (tools/ring_bench, built beforehand by tools/ring_dbg.sh build) and the streaming loop's parts
[ -x $R/tools/ring_bench ] && (cd $R && tools/ring_bench 0 7 1 > $OUT/ring_microbench.txt 2>&1)
XP_BENCH_PCIE_PARTS=1 python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-backend 2>&1 >/dev/null | grep "pcie parts" > $OUT/streaming_parts.txt
# the headline line again with THIS run's PMC traffic in roofline.traffic (bench.py reads profiles/pmc_traffic.json)
cp $OUT/bench.json $OUT/bench_before_pmc.json
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json
python3 $R/bench.py > $OUT/bench.json 2> $OUT/hip_event_breakdown.txt
python3 $R/bench.py --config c4 --steps 30 --no-cpu-baseline > $OUT/c4_bench.json 2> $OUT/c4_hip_event_breakdown.txt
python3 $R/bench.py --config c5 --steps 30 --no-cpu-baseline > $OUT/c5_bench.json 2> $OUT/c5_hip_event_breakdown.txt
python3 $R/bench.py --config c5 --pairs 32 --steps 8 --no-cpu-baseline > $OUT/c5_32pairs_bench.json 2> /dev/null
cat $OUT/bench.json; head -12 $OUT/kernel_stats.csv | cut -c1-200; head -20 $OUT/pmc_hbm_traffic.txt
