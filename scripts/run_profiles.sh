#!/bin/bash
# usage (GPU box): scripts/run_profiles.sh TAG  -> gpurun_out/prof_TAG_{stats,fetch,write,sq}; kernel stats + PMC of the default bench
TAG=$1
export PYTHONPATH=$GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_stats -- python3 $ARGS > $R/gpurun_out/prof_${TAG}_stats.log 2>&1
PARGS="$R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $PARGS > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $PARGS > $R/gpurun_out/prof_${TAG}_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/prof_${TAG}_sq -- python3 $PARGS > $R/gpurun_out/prof_${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof_${TAG}_sq2 -- python3 $PARGS > $R/gpurun_out/prof_${TAG}_sq2.log 2>&1
python3 $R/scripts/pmc_summary.py --kernel "scan_kernel|scan_wide_kernel|scan_two_rows_kernel|scan_noadj_kernel|init_kernel|pairs_collect|read_probe" $R/gpurun_out/prof_${TAG}_fetch $R/gpurun_out/prof_${TAG}_write $R/gpurun_out/prof_${TAG}_sq $R/gpurun_out/prof_${TAG}_sq2 > $R/gpurun_out/prof_${TAG}_pmc.txt 2>&1
find $R/gpurun_out/prof_${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_${TAG}_kernel_stats.csv \;
