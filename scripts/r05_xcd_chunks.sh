#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
AB_ARGS="--shape 1" bash scripts/ab_variants.sh gpurun_out/r05l_ab_wide.txt base xc2 xc8 xc16 xc32 xc64 base > /dev/null
cut -c1-100 gpurun_out/r05l_ab_wide.txt
for v in base xc8 xc16 xc32 xc64; do
  if [ "$v" = base ]; then unset TISSUE_SCAN_LIB; else export TISSUE_SCAN_LIB=$R/scratch/lib$v.so; fi
  D=$R/gpurun_out/r05l_fetch_$v
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D -- python3 $R/scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 4 --no-check --shape 1 > $D.log 2>&1)
  echo "## $v" >> gpurun_out/r05l_fetch.txt
  python3 scripts/pmc_summary.py --kernel "scan_wide_kernel" $D | grep -v "^==" >> gpurun_out/r05l_fetch.txt
  rm -rf $D
done
cut -c1-150 gpurun_out/r05l_fetch.txt
