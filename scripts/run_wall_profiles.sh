#!/bin/bash
# usage (GPU box): scripts/run_wall_profiles.sh TAG [CONFIG] -> gpurun_out/wall_TAG_*: probe output, kernel stats and SQ counters of the
# wall-voxel kernels on one configuration (default C2)
TAG=$1
CFG=${2:-C2}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
export TA_WALL_VERBOSE=1
timeout -k 10 200 python3 $R/scripts/probe_walls.py $CFG > $R/gpurun_out/wall_${TAG}_probe.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wall_${TAG}_stats -- python3 $R/scripts/probe_walls.py $CFG > /dev/null 2>&1 || exit 1
find $R/gpurun_out/wall_${TAG}_stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/wall_${TAG}_kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/wall_${TAG}_sq -- python3 $R/scripts/probe_walls.py $CFG > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/wall_${TAG}_sq2 -- python3 $R/scripts/probe_walls.py $CFG > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/wall_${TAG}_fetch -- python3 $R/scripts/probe_walls.py $CFG > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/wall_${TAG}_write -- python3 $R/scripts/probe_walls.py $CFG > /dev/null 2>&1
python3 $R/scripts/pmc_summary.py --kernel "wall_cells|wall_copy" $R/gpurun_out/wall_${TAG}_sq $R/gpurun_out/wall_${TAG}_sq2 $R/gpurun_out/wall_${TAG}_fetch $R/gpurun_out/wall_${TAG}_write > $R/gpurun_out/wall_${TAG}_pmc.txt 2>&1
