#!/bin/bash
# usage (GPU box): scripts/r05_ladder.sh TAG   -- the ablation ladder, record counts, stamps and PMC of the WIDE sweep kernel
TAG=$1
R=$GRAFT_REPO_ROOT
cd $R
export PYTHONPATH=$R
export AB_ARGS="--shape 1"
bash scripts/ab_variants.sh gpurun_out/${TAG}_ladder.txt base noflush hot1 hot2 hot4 nohotflush abl1 abl3 nf01 base hot1f hot2f share1 noboxhot nofp nofl mom4 > /dev/null
for v in reccount stamps; do
  TISSUE_SCAN_LIB=$R/scratch/lib$v.so python3 scripts/probe_stamps.py C4 --shape=1 >> gpurun_out/${TAG}_stamps.txt 2>&1
  TISSUE_SCAN_LIB=$R/scratch/lib$v.so python3 scripts/probe_stamps.py C4 --shape=1 --no-ellipsoid 2>&1 | sed 's/^/filled /' >> gpurun_out/${TAG}_stamps.txt
done
bash scripts/run_pmc_ablations.sh ${TAG} --shape 1 -- base abl1 nf01 abl3 > /dev/null 2>&1
tail -40 gpurun_out/${TAG}_ladder.txt
