#!/bin/bash
# usage (GPU box): scripts/ab_wallsort.sh OUT.txt name1 name2 ...  ("base" = the product library): the grouped wall-voxel
# fetch of C2 and C3 under each variant library scratch/libNAME.so (scripts/build_variant.py NAME -DTA_RS_...)
OUT=$1; shift
export PYTHONPATH=$PWD
for v in "$@"; do
  if [ "$v" = base ]; then unset TISSUE_SCAN_LIB; else export TISSUE_SCAN_LIB=$PWD/scratch/lib$v.so; fi
  echo "== $v" >> $OUT
  python3 scripts/probe_walls.py C2 2>&1 | grep "grouped" | cut -c1-60 >> $OUT
  python3 scripts/probe_walls.py C3 2>&1 | grep "grouped" | sed 's/^/C3 /' | cut -c1-64 >> $OUT
done
cat $OUT
