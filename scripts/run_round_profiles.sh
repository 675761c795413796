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
# ablation builds (scripts/build_variant.py NAME -D...; results wrong by construction, only the time matters), a CUMULATIVE ladder
# on the WIDE shape (--shape 1: the kernel the headline names):
#   noflush = the tile tables are not flushed; hot1 = ... and the top-of-plane drains add nothing to the tables; hot2 = ... and run
#   no probe rounds; hot4 = ... and read neither keys nor records; noslow = ... and the in-plane drains / the end of the tile consume
#   nothing; abl1 = nothing is consumed anywhere (records produced and stored); abl3 = not even placed (compares + counts + scan);
#   nf01 (not cumulative) = no axis-0 / axis-1 face records; nomask = the pre-round-5 record stores (trash slot, five VALU a position);
#   nohalo1 = a workgroup's first wave reads no row above (12.5 % fewer bytes fetched); notr = the flush adds a label's sums lane = label
OUT=gpurun_out/${TAG}_ablations.txt
: > $OUT
for a in base noflush hot1 hot2 hot4 noslow abl1 abl3 nf01 nomask nohalo1 notr base; do
  if [ "$a" = base ] || [ -f scratch/lib$a.so ]; then
    echo "ablation $a" >> $OUT
    if [ "$a" = base ]; then unset TISSUE_SCAN_LIB; else export TISSUE_SCAN_LIB=$R/scratch/lib$a.so; fi
    python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 7 --no-check --shape 1 2>&1 | grep "impl=0" | cut -c1-110 >> $OUT
    python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 5 --no-check --no-ellipsoid --shape 1 2>&1 | grep "impl=0" | sed 's/^/tissue-filled /' | cut -c1-124 >> $OUT
  fi
done
unset TISSUE_SCAN_LIB
if [ -f scratch/libreccount.so ]; then
  echo "records per launch, wide shape (flags 8..12: faces, runs, drains through the in-plane drain_buffers; faces, runs through the top-of-plane drains)" >> $OUT
  TISSUE_SCAN_LIB=$R/scratch/libreccount.so python3 scripts/probe_stamps.py C4 --shape=1 2>&1 | grep "counters" >> $OUT
  TISSUE_SCAN_LIB=$R/scratch/libreccount.so python3 scripts/probe_stamps.py C4 --shape=1 --no-ellipsoid 2>&1 | grep "counters" | sed 's/^/tissue-filled /' >> $OUT
fi
if [ -f scratch/libbarstamp.so ]; then
  echo "a workgroup's life, wide shape (flags 8..11: cycles >> 8 summed over the waves: sweep of the tile, wait at the barrier before the flush, flush; wave-tiles)" >> $OUT
  TISSUE_SCAN_LIB=$R/scratch/libbarstamp.so python3 scripts/probe_stamps.py C4 --shape=1 2>&1 | grep "counters" >> $OUT
  TISSUE_SCAN_LIB=$R/scratch/libbarstamp.so python3 scripts/probe_stamps.py C4 --shape=1 --no-ellipsoid 2>&1 | grep "counters" | sed 's/^/tissue-filled /' >> $OUT
fi
echo "full" >> $OUT
python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f 0x17 0x0f 0x07 --iters 7 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> $OUT
python3 scripts/probe_impls.py C5 --impl 0 --feat 0x1f --iters 4 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> $OUT
python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f 0x0f --iters 5 --no-check --no-ellipsoid 2>&1 | grep "impl=0" | sed 's/^/tissue-filled /' | cut -c1-124 >> $OUT
python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 5 --no-check --no-ellipsoid --shape 0 2>&1 | grep "impl=0" | sed 's/^/tissue-filled narrow /' | cut -c1-124 >> $OUT
python3 scripts/probe_impls.py C4 --impl 0 --feat 0x1f --iters 5 --no-check --shape 0 2>&1 | grep "impl=0" | sed 's/^/C4 narrow /' | cut -c1-124 >> $OUT
python3 scripts/probe_impls.py C2 --dims 1024 1024 1024 --impl 0 --feat 0x1f 0x0f --iters 5 --no-check 2>&1 | grep "impl=0" | sed 's/^/1024^3 uint16 /' | cut -c1-124 >> $OUT
python3 scripts/probe_impls.py C2 --impl 0 --feat 0x07 0x1f --iters 15 --no-check 2>&1 | grep "impl=0" | cut -c1-110 >> $OUT
python3 scripts/probe_impls.py C4 --dims 1000 1000 1000 --impl 0 --feat 0x1f 0x0f --iters 7 --no-check 2>&1 | grep "impl=0" | sed 's/^/1000^3 /' | cut -c1-118 >> $OUT
python3 scripts/probe_walls.py C2 2>&1 | tail -2 >> $OUT

# PMC passes of the wide kernel and its abl1 / nf01 builds on C4 and on the tissue-filled C4 (instruction counts, LDS, waits)
bash scripts/run_pmc_ablations.sh ${TAG} --shape 1 -- base abl1 nf01 > /dev/null 2>&1
bash scripts/run_pmc_ablations.sh ${TAG}f --shape 1 --no-ellipsoid -- base abl1 > /dev/null 2>&1
# the wall-voxel passes on C2: kernel stats of two plain + two grouped fetches (scripts/prof_walls.sh), and the record sort of round 4
# (TA_WALL_KEYED=0: the plain fetch's records sorted, a key pass before and a coordinate gather behind) in the same call
timeout -k 10 300 bash scripts/prof_walls.sh ${TAG} > gpurun_out/${TAG}_walls_summary.txt 2>&1
python3 scripts/probe_walls.py C2 2>&1 | grep "grouped\|records" >> gpurun_out/${TAG}_walls_summary.txt
TA_WALL_KEYED=0 python3 scripts/probe_walls.py C2 2>&1 | grep "grouped" | sed 's/^/TA_WALL_KEYED=0 /' >> gpurun_out/${TAG}_walls_summary.txt
python3 scripts/probe_walls.py C3 2>&1 | grep "grouped\|records" >> gpurun_out/${TAG}_walls_summary.txt
cd $R
if [ -f scratch/pipes_bench ]; then ./scratch/pipes_bench > gpurun_out/${TAG}_pipes.txt 2>&1; fi
python3 scripts/update_pmc_traffic.py C4 gpurun_out/prof_${TAG}_pmc.txt --note "default bench" > gpurun_out/${TAG}_traffic.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
