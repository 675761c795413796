#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
for v in prev base; do
  if [ "$v" = base ]; then unset TISSUE_SCAN_LIB; else export TISSUE_SCAN_LIB=$R/scratch/lib$v.so; fi
  echo "== $v"
  python3 scripts/probe_impls.py C2 --impl 0 --feat 0x1f 0x07 --iters 15 --no-check 2>&1 | grep "impl=0" | cut -c1-100
  python3 scripts/probe_impls.py C2 --impl 0 --feat 0x1f --iters 9 --no-check --no-ellipsoid 2>&1 | grep "impl=0" | sed 's/^/filled /' | cut -c1-107
  python3 scripts/probe_impls.py C2 --dims 1024 1024 1024 --impl 0 --feat 0x1f 0x0f --iters 5 --no-check 2>&1 | grep "impl=0" | sed 's/^/1024^3 u16 /' | cut -c1-112
  python3 scripts/probe_impls.py C5 --impl 0 --feat 0x1f --iters 4 --no-check 2>&1 | grep "impl=0" | sed 's/^/C5 /' | cut -c1-104
  python3 scripts/probe_impls.py C4 --dims 1000 1000 1000 --impl 0 --feat 0x1f --iters 5 --no-check 2>&1 | grep "impl=0" | sed 's/^/1000^3 /' | cut -c1-108
done
