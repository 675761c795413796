#!/bin/bash
# usage (GPU box): scripts/prof_walls.sh TAG [env assignments...]  -- rocprofv3 kernel stats of the wall-voxel passes on C2 (two plain + two grouped fetches)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_walls -- python3 $R/scripts/probe_walls.py C2 > $R/gpurun_out/prof_${TAG}_walls.log 2>&1
cd $R
find gpurun_out/prof_${TAG}_walls -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_walls_kernel_stats.csv \;
test -f gpurun_out/${TAG}_walls_kernel_stats.csv && python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_walls_kernel_stats.csv")))
for r in rows[:16]:
    print("%-70s calls %3s avg %9.1f us total %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
tail -4 gpurun_out/prof_${TAG}_walls.log
