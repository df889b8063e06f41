#!/bin/bash
# Stall counters of the split-fp16 GEMM (gemm_h2_kernel) on the deep-stage shapes — VERDICT r2 next 3(a).
# One shape per process (GB_ONLY), counters in their own passes (--pmc + --kernel-trace only), program directly after `--`.
#   usage (GPU box, repo root): bash tools/h2_stalls.sh   ->  gpurun_out/h2_stalls.txt
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/h2_stalls; rm -rf $OUT; mkdir -p $OUT
export GB_H2=1
SUM=$R/gpurun_out/h2_stalls.txt; : > $SUM
for shape in ${H2_SHAPES:-10 12 15 13}; do     # gemm_bench.py indices: 10 = M19200 N384 K384, 12 = M19200 N384 K1536, 15 = M4800 N768 K3072, 13 = M4800 N768 K768
  export GB_ONLY=$shape
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES" \
             "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_VALU" \
             "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/s$shape/p$i -- python3 $R/tools/gemm_bench.py > $OUT/s$shape.p$i.log 2>&1 || echo "shape $shape pass $i failed: $(tail -2 $OUT/s$shape.p$i.log)" >> $SUM
  done
  echo "=== shape index $shape: $(grep -h '^M' $OUT/s$shape.p1.log | head -1)   (time under the profiler)" >> $SUM
  python3 $R/tools/pmc_generic.py $OUT/s$shape gemm_h2 >> $SUM 2>&1
done
unset GB_ONLY
echo "=== un-profiled microbench of the same shapes" >> $SUM
GB_ONLY=10,12,15,13 python3 $R/tools/gemm_bench.py 2>/dev/null | grep '^M' >> $SUM
find $OUT -name "*.csv" -size +2000k -delete
cat $SUM
