#!/bin/bash
# SQ counters of the NT kernel in one arithmetic mode.  usage: tools/pmc_x3p.sh MODE K
MODE=$1; K=$2
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $O
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
         "SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS" \
         "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_CYCLES"; do
  d=/tmp/pmcx_${MODE}_$i; rm -rf $d
  ( cd $GRAFT_REPO_ROOT && GEMM_MODE=$MODE timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- python tools/gemm_bench.py 96000 $K 300 5 > $O/pmcx_${MODE}_$i.log 2>&1 )
  python $GRAFT_REPO_ROOT/tools/pmc_sum.py $d | awk '/^==/{p=index($0,"gemm_nt_kernel")>0} p'
  i=$((i+1))
done > $O/pmcx_${MODE}_$K.txt
cat $O/pmcx_${MODE}_$K.txt
