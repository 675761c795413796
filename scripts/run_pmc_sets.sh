#!/bin/bash
# usage (GPU box): scripts/run_pmc_sets.sh TAG "COUNTERS SET 1" ["COUNTERS SET 2" ...] -- name1 name2 ...   ("base" = the product library)
# one rocprofv3 --pmc run per counter set and variant library scratch/libNAME.so on C4 (mask 0x1f); summary -> gpurun_out/pmcsets_TAG.txt
TAG=$1; shift
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmcsets_${TAG}.txt
: > $OUT
for v in "$@"; do
  if [ "$v" = base ]; then unset TISSUE_SCAN_LIB; else export TISSUE_SCAN_LIB=$R/scratch/lib$v.so; fi
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    D=$R/gpurun_out/pmcsets_${TAG}_${v}_$i
    rocprofv3 --pmc $set --output-format csv -d $D -- python3 $R/scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 4 --no-check $PMC_PROBE_ARGS > $D.log 2>&1 || echo "$v pass $i failed" >> $OUT
    echo "## $v" >> $OUT
    python3 $R/scripts/pmc_summary.py --kernel "scan_two_rows_kernel|scan_kernel|scan_wide_kernel" $D | grep -v "^==" >> $OUT 2>&1
    rm -rf $D
  done
done
cat $OUT
