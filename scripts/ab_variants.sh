#!/bin/bash
# usage (GPU box): scripts/ab_variants.sh OUT.txt name1 name2 ...   ("base" = the product library)
# times C4 and the tissue-filled C4 (mask 0x1f) under each variant library scratch/libNAME.so, with the parity check of
# probe_impls.py switched off; AB_ARGS="--shape 1" forces the tile shape;
# probe_impls.py switched off (ablations are wrong by construction; real candidates go through the test suite)
OUT=$1; shift
export PYTHONPATH=$PWD
for v in "$@"; do
  if [ "$v" = base ]; then unset TISSUE_SCAN_LIB; else export TISSUE_SCAN_LIB=$PWD/scratch/lib$v.so; fi
  echo "== $v" >> $OUT
  python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 7 --no-check $AB_ARGS 2>&1 | grep "impl=0\|Error\|error" | cut -c1-160 >> $OUT
  python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 5 --no-check --no-ellipsoid $AB_ARGS 2>&1 | grep "impl=0\|Error\|error" | sed 's/^/filled /' | cut -c1-128 >> $OUT
done
cat $OUT
