#!/bin/bash
# SQ counter passes over a micro-benchmark command, aggregated per kernel.  usage: tools/pmc_kernel.sh TAG FILTER -- cmd...
TAG=$1; FILTER=$2; shift 3
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC" \
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  d=/tmp/pmck_${TAG}_$i; rm -rf $d
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- "$@" > $O/pmck_${TAG}_$i.log 2>&1 )
  python $GRAFT_REPO_ROOT/tools/pmc_sum.py $d | awk -v f="$FILTER" '/^==/{p=index($0,f)>0} p' 
  i=$((i+1))
done > $O/pmck_$TAG.txt
cat $O/pmck_$TAG.txt
