#!/bin/bash
# usage (GPU box): scripts/run_round_profiles.sh TAG
# everything profiles/ holds for a round: kernel stats + PMC of the default bench (run_profiles.sh), the moments-only
# mask, the uint16 kernels, the bench lines of the other configurations, the ablation ladder (scratch/lib*.so when present).
# Every step appends to a file under gpurun_out/ (a silent GPU call is taken to be hung after 7 minutes).
TAG=$1
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
bash $R/scripts/run_profiles.sh $TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_mom_stats -- python3 $R/bench.py --features 0x0f --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_${TAG}_mom_stats.log 2>&1
find $R/gpurun_out/prof_${TAG}_mom_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_mom_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_mom_fetch -- python3 $R/bench.py --features 0x0f --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_${TAG}_mom_fetch.log 2>&1
python3 $R/scripts/pmc_summary.py --kernel "scan_noadj_kernel" $R/gpurun_out/prof_${TAG}_mom_fetch > $R/gpurun_out/prof_${TAG}_mom_pmc.txt 2>&1
# the uint16 kernels (C2: 512^3 uint16): its own mask 0x07 (scan_noadj_kernel<ushort>) and the full feature set (scan_kernel<ushort>)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_u16_stats -- python3 $R/bench.py --config C2 --features 0x1f --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_${TAG}_u16_stats.log 2>&1
find $R/gpurun_out/prof_${TAG}_u16_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_u16_kernel_stats.csv \;
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/prof_${TAG}_u16_sq -- python3 $R/bench.py --config C2 --features 0x1f --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_${TAG}_u16_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_u16_fetch -- python3 $R/bench.py --config C2 --features 0x1f --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_${TAG}_u16_fetch.log 2>&1
python3 $R/scripts/pmc_summary.py --kernel "scan_kernel|scan_noadj_kernel" $R/gpurun_out/prof_${TAG}_u16_sq $R/gpurun_out/prof_${TAG}_u16_fetch > $R/gpurun_out/prof_${TAG}_u16_pmc.txt 2>&1
cd $R
: > gpurun_out/${TAG}_bench.jsonl
python3 bench.py 2>gpurun_out/${TAG}_bench_default.err | tail -1 >> gpurun_out/${TAG}_bench.jsonl
for extra in "--config C3" "--config C2" "--config C2 --features 0x1f" "--config C1 --features 0x1f" "--features 0x0f" "--features 0x07" "--dims 1000 1000 1000"; do
  python3 bench.py $extra --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 >> gpurun_out/${TAG}_bench.jsonl
done
# ablation builds (scripts/build_variant.py NAME -D...; results wrong by construction, only the time matters):
#   abl1 = records produced, not consumed; abl3 = not even placed; nf01 = no axis-0 / axis-1 face records;
#   noflush = the tile tables are not flushed; hot1 / hot2 / hot4 (all without flush) = the top-of-plane drains add nothing to
#   the tables / ... and run no probe rounds / ... and read no records; nohotflush = no top-of-plane drains at all, no flush
OUT=gpurun_out/${TAG}_ablations.txt
: > $OUT
for a in abl1 abl3 nf01 noflush hot1 hot2 hot4 nohotflush; do
  if [ -f scratch/lib$a.so ]; then
    echo "ablation $a" >> $OUT
    TISSUE_SCAN_LIB=$R/scratch/lib$a.so python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 7 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> $OUT
    TISSUE_SCAN_LIB=$R/scratch/lib$a.so python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 5 --no-check --no-ellipsoid 2>&1 | grep "impl=0" | sed 's/^/tissue-filled /' | cut -c1-124 >> $OUT
  fi
done
if [ -f scratch/libreccount.so ]; then
  echo "records per launch (flags 8..12: faces, runs, drains through drain_buffers; faces, runs through the top-of-plane drains)" >> $OUT
  TISSUE_SCAN_LIB=$R/scratch/libreccount.so python3 scripts/probe_stamps.py C4 2>&1 | grep "counters" >> $OUT
  TISSUE_SCAN_LIB=$R/scratch/libreccount.so python3 scripts/probe_stamps.py C4 --no-ellipsoid 2>&1 | grep "counters" | sed 's/^/tissue-filled /' >> $OUT
fi
echo "full" >> $OUT
python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f 0x17 0x0f 0x07 --iters 7 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> $OUT
python3 scripts/probe_impls.py C5 --impl 0 --feat 0x1f --iters 4 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> $OUT
python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f 0x0f --iters 5 --no-check --no-ellipsoid 2>&1 | grep "impl=0" | sed 's/^/tissue-filled /' | cut -c1-124 >> $OUT
python3 scripts/probe_impls.py C2 --impl 0 --feat 0x07 0x1f --iters 15 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> $OUT
python3 scripts/probe_impls.py C4 --dims 1000 1000 1000 --impl 0 --feat 0x1f 0x0f --iters 7 --no-check 2>&1 | grep "impl=0" | sed 's/^/1000^3 /' | cut -c1-118 >> $OUT
python3 scripts/probe_walls.py C2 2>&1 | tail -2 >> $OUT
