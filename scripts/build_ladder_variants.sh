#!/bin/bash
# the ablation / instrumentation builds scripts/run_round_profiles.sh times (switches: tissue_analysis_amd/csrc/ta_sweep_switches.h)
cd "$(dirname "$0")/.."
mkdir -p scratch
bash scripts/build_variants.sh "abl1 -DTA_ABLATE=1" "abl3 -DTA_ABLATE=3" "nf01 -DTA_ABL_NOFACE0 -DTA_ABL_NOFACE1" \
  "noflush -DTA_ABL_NOFLUSH" "hot1 -DTA_ABL_NOFLUSH -DTA_ABL_HOT=1" "hot2 -DTA_ABL_NOFLUSH -DTA_ABL_HOT=2" \
  "hot4 -DTA_ABL_NOFLUSH -DTA_ABL_HOT=4" "noslow -DTA_ABL_NOFLUSH -DTA_ABL_HOT=4 -DTA_ABL_NOSLOW" \
  "reccount -DTA_RECCOUNT" "barstamp -DTA_BARSTAMP -DTA_ABL_X" "nomask -DTA_MASKED_STORES=0" \
  "nohalo1 -DTA_ABL_NOHALO=1" "notr -DTA_FLUSH_TRANSPOSE=0"
hipcc --offload-arch=gfx950 -O2 scripts/pipes_bench.hip -o scratch/pipes_bench 2>/dev/null || echo "pipes_bench: build failed"
