#!/bin/bash
# PMC passes on the fused-tail microbench (tools/mlp_bench.py, split-fp16 instances): where do the wave cycles of mlp_fused_kernel go?
# usage on the GPU box: bash tools/mlp_pmc.sh
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/mlp_pmc; rm -rf $OUT; mkdir -p $OUT
export MLP_H2=1 MLP_FUSED_ONLY=1 MLP_ONLY=0,1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/mlp_bench.py > $OUT/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $OUT/p$i.log)"
done
python3 $R/tools/pmc_generic.py $OUT mlp_fused > $R/gpurun_out/mlp_pmc_summary.txt 2>&1
find $OUT -name "*.csv" -size +2000k -delete
head -150 $R/gpurun_out/mlp_pmc_summary.txt
