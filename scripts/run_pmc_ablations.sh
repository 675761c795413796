#!/bin/bash
# usage (GPU box): scripts/run_pmc_ablations.sh TAG [--no-ellipsoid] -- name1 name2 ...   ("base" = the product library)
# PMC passes (instruction counts, wait / active cycles, LDS) of the sweep kernel under each variant library
# scratch/libNAME.so on C4, one rocprofv3 run per counter set; summary -> gpurun_out/pmcabl_TAG.txt
# (differences between the full build and abl1 / nf01 / abl3 = what the drain / the face records / the placement cost)
TAG=$1; shift
EXTRA=""
while [ "$1" != "--" ]; do EXTRA="$EXTRA $1"; shift; done
shift
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmcabl_${TAG}.txt
: > $OUT
for v in "$@"; do
  if [ "$v" = base ]; then unset TISSUE_SCAN_LIB; else export TISSUE_SCAN_LIB=$R/scratch/lib$v.so; fi
  i=0
  for set in \
   "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
   "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC" \
   "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES" ; do
    i=$((i+1))
    D=$R/gpurun_out/pmcabl_${TAG}_${v}_$i
    rocprofv3 --pmc $set --output-format csv -d $D -- python3 $R/scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 4 --no-check $EXTRA > $D.log 2>&1 || echo "$v pass $i failed" >> $OUT
    echo "## $v" >> $OUT
    python3 $R/scripts/pmc_summary.py --kernel "scan_two_rows_kernel|scan_kernel|scan_wide_kernel" $D | grep -v "^==" >> $OUT 2>&1
    rm -rf $D
  done
done
cat $OUT
