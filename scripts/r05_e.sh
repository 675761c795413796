#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
./scratch/pipes_bench > gpurun_out/r05_pipes.txt 2>&1
for v in reccount fd1rc; do
  TISSUE_SCAN_LIB=$R/scratch/lib$v.so python3 scripts/probe_stamps.py C4 --shape=1 2>&1 | grep "counters" | sed "s/^/$v /" >> gpurun_out/r05e_rc.txt
  TISSUE_SCAN_LIB=$R/scratch/lib$v.so python3 scripts/probe_stamps.py C4 --shape=1 --no-ellipsoid 2>&1 | grep "counters" | sed "s/^/$v filled /" >> gpurun_out/r05e_rc.txt
done
AB_ARGS="--shape 1" bash scripts/ab_variants.sh gpurun_out/r05e_ab_wide.txt base fd1 base fd1 > /dev/null
PMC_PROBE_ARGS="--shape 1 --no-ellipsoid" bash scripts/run_pmc_sets.sh r05e "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY" -- base fd1 > /dev/null 2>&1
cat gpurun_out/r05_pipes.txt; cat gpurun_out/r05e_rc.txt | cut -c1-250; cut -c1-100 gpurun_out/r05e_ab_wide.txt; grep -v "^==" gpurun_out/pmcsets_r05e.txt | grep "scan_wide\|##" | cut -c55-140
