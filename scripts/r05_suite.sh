#!/bin/bash
# usage (GPU box): scripts/r05_suite.sh TAG  -- the whole GPU suite, then the default bench line
TAG=$1
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_suite.log 2>&1; echo "suite rc=$?" >> gpurun_out/${TAG}_suite.log
tail -5 gpurun_out/${TAG}_suite.log
python3 bench.py 2> gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench.json
python3 - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json"))
r=d["roofline"]; s=d.get("secondary",{})
print("C4 kernel_ms", r["kernel_ms"], "frac", r["frac"], "kernel", r["kernel"], "ms_per_step", d["ms_per_step"], "frac50k", r.get("frac_50k_labels"))
for k in ("tissue_filled","c5_single_gpu","wall_voxels_c2","sparse_ids","two_steps_in_flight"):
    print(k, {a:b for a,b in s.get(k,{}).items() if a in ("kernel_ms","roofline_frac","ms_per_step","kernels_ms","census_ms","rank_copy_ms","skipped")})
PY
