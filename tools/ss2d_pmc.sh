#!/bin/bash
# PMC passes on the fused SS2D core microbench (tools/ss2d_bench.py, both precision classes): where do the wave cycles of ss2d_pass1 / pass3 / seq_scan go?
# (the VALU-bound claim was resting on a round-1 file.)   usage on the GPU box: bash tools/ss2d_pmc.sh
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ss2d_pmc; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/ss2d_bench.py 0 1 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python3 $R/tools/pmc_generic.py $OUT ss2d > $R/gpurun_out/ss2d_pmc_summary.txt 2>&1
find $OUT -name "*.csv" -size +2000k -delete
head -120 $R/gpurun_out/ss2d_pmc_summary.txt
