#!/bin/bash
# usage (GPU box): scripts/run_round_profiles.sh TAG
# everything profiles/ holds for a round: kernel stats + PMC of the default bench (run_profiles.sh), the moments-only
# mask, the bench lines of the other configurations, the ablation builds (scratch/libabl{1,2,3}.so when present).
TAG=$1
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
bash $R/scripts/run_profiles.sh $TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_mom_stats -- python3 $R/bench.py --features 0x0f --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_${TAG}_mom_stats.log 2>&1
find $R/gpurun_out/prof_${TAG}_mom_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_mom_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_mom_fetch -- python3 $R/bench.py --features 0x0f --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_${TAG}_mom_fetch.log 2>&1
python3 $R/scripts/pmc_summary.py --kernel "scan_noadj_kernel" $R/gpurun_out/prof_${TAG}_mom_fetch > $R/gpurun_out/prof_${TAG}_mom_pmc.txt 2>&1
cd $R
: > gpurun_out/${TAG}_bench.jsonl
python3 bench.py 2>gpurun_out/${TAG}_bench_default.err | tail -1 >> gpurun_out/${TAG}_bench.jsonl
for extra in "--config C3" "--config C2" "--features 0x0f" "--features 0x07"; do
  python3 bench.py $extra --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 >> gpurun_out/${TAG}_bench.jsonl
done
# ablation builds (scripts/build_variant.py NAME -D...; results wrong by construction, only the time matters):
#   abl1 = records produced, not consumed; abl2 = not stored; abl3 = not even placed; nf0 / nf1 / nf01 = no axis-0 / axis-1 / neither
#   face records; l2 = every plane re-reads the tile's first plane (the volume comes from L2: what is left is not HBM time);
#   reccount = the full kernel counting its records (slow: the counters are global atomics)
for a in abl1 abl2 abl3 nf0 nf1 nf01 l2; do
  if [ -f scratch/lib$a.so ]; then
    echo "ablation $a" >> gpurun_out/${TAG}_ablations.txt
    TISSUE_SCAN_LIB=$R/scratch/lib$a.so python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 7 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> gpurun_out/${TAG}_ablations.txt
    TISSUE_SCAN_LIB=$R/scratch/lib$a.so python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 5 --no-check --no-ellipsoid 2>&1 | grep "impl=0" | sed 's/^/tissue-filled /' | cut -c1-124 >> gpurun_out/${TAG}_ablations.txt
  fi
done
if [ -f scratch/libreccount.so ]; then
  echo "records per launch" >> gpurun_out/${TAG}_ablations.txt
  TISSUE_SCAN_LIB=$R/scratch/libreccount.so python3 scripts/probe_stamps.py C4 2>&1 | grep "records per launch" >> gpurun_out/${TAG}_ablations.txt
  TISSUE_SCAN_LIB=$R/scratch/libreccount.so python3 scripts/probe_stamps.py C4 --no-ellipsoid 2>&1 | grep "records per launch" | sed 's/^/tissue-filled /' >> gpurun_out/${TAG}_ablations.txt
fi
echo "full" >> gpurun_out/${TAG}_ablations.txt
python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f 0x17 0x0f 0x07 --iters 7 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> gpurun_out/${TAG}_ablations.txt
python3 scripts/probe_impls.py C5 --impl 0 --feat 0x1f --iters 4 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> gpurun_out/${TAG}_ablations.txt
python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f 0x0f --iters 5 --no-check --no-ellipsoid 2>&1 | grep "impl=0" | sed 's/^/tissue-filled /' | cut -c1-124 >> gpurun_out/${TAG}_ablations.txt
python3 scripts/probe_impls.py C2 --impl 0 --feat 0x07 --iters 15 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> gpurun_out/${TAG}_ablations.txt
python3 scripts/probe_walls.py C2 2>&1 | tail -2 >> gpurun_out/${TAG}_ablations.txt
