#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
for a in "" "--no-ellipsoid"; do
  TISSUE_SCAN_LIB=$R/scratch/libbarstamp.so python3 scripts/probe_stamps.py C4 --shape=1 $a 2>&1 | grep counters | sed "s/^/barstamp $a /" | cut -c1-300
done
AB_ARGS="--shape 1" bash scripts/ab_variants.sh gpurun_out/r05i_ab_wide.txt r1c0 w2 w2b noslow r1c0 > /dev/null
cut -c1-100 gpurun_out/r05i_ab_wide.txt
