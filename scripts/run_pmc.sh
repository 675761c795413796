#!/bin/bash
# usage: scripts/run_pmc.sh TAG "probe args"   (on the GPU box; writes gpurun_out/pmc_TAG_*)
TAG=$1; shift
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_IFETCH" \
 "SQ_INST_LEVEL_SMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_CYCLES" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/scripts/probe_impls.py "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_${TAG}_* > $R/gpurun_out/pmc_${TAG}_summary.txt 2>&1
