#!/bin/bash
# usage: scripts/build_variants.sh "name1 -DFLAG ..." "name2 -D..." ...   -> scratch/libNAME.so, three builds at a time
# (ablation builds: results wrong by construction -> --no-pin-check is added when the spec contains ABL)
run() {
  spec="$1"; name=${spec%% *}
  extra=""; case "$spec" in *ABL*|*DBG*) extra="--no-pin-check";; esac
  python3 scripts/build_variant.py $spec $extra > scratch/build_$name.log 2>&1 || { echo "FAILED $name"; tail -5 scratch/build_$name.log; }
}
n=0
for spec in "$@"; do
  run "$spec" &
  n=$((n+1)); if [ $((n % 3)) = 0 ]; then wait; fi
done
wait
ls -la scratch/lib*.so | awk '{print $6, $7, $8, $9}'
