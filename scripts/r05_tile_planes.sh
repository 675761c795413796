#!/bin/bash
# usage (GPU box): scripts/r05_tile_planes.sh OUT  -- the sweep over the tile height of each adjacency kernel (the first line of a run is the warm-up)
OUT=$1
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
P="python3 scripts/probe_impls.py"
for i in 1 2; do
  $P C4 --impl 0 --feat 0x1f --tp 32 24 28 32 --iters 9 --no-check --shape 1 2>&1 | grep "impl=0" | sed "s/^/C4 wide   /" | cut -c1-100 >> $OUT
  $P C4 --impl 0 --feat 0x1f --tp 40 32 40 48 --iters 9 --no-check --shape 0 2>&1 | grep "impl=0" | sed "s/^/C4 narrow /" | cut -c1-100 >> $OUT
  $P C4 --impl 0 --feat 0x1f --tp 32 16 24 32 --iters 7 --no-check --shape 1 --no-ellipsoid 2>&1 | grep "impl=0" | sed "s/^/filled wide   /" | cut -c1-104 >> $OUT
  $P C4 --impl 0 --feat 0x1f --tp 40 24 32 40 48 --iters 7 --no-check --shape 0 --no-ellipsoid 2>&1 | grep "impl=0" | sed "s/^/filled narrow /" | cut -c1-104 >> $OUT
  $P C2 --dims 1024 1024 1024 --impl 0 --feat 0x1f --tp 32 16 24 28 32 --iters 7 --no-check 2>&1 | grep "impl=0" | sed "s/^/1024^3 uint16 /" | cut -c1-104 >> $OUT
done
$P C5 --impl 0 --feat 0x1f --tp 32 24 28 32 --iters 4 --no-check 2>&1 | grep "impl=0" | sed "s/^/C5 wide /" | cut -c1-100 >> $OUT
cat $OUT
