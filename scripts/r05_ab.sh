#!/bin/bash
# usage (GPU box): scripts/r05_ab.sh TAG "pytest args or empty" name1 name2 ...  -- parity tests first, then same-call A/B of variant libraries (both shapes)
TAG=$1; shift
TESTS=$1; shift
R=$GRAFT_REPO_ROOT
cd $R
export PYTHONPATH=$R
if [ -n "$TESTS" ]; then
  timeout -k 10 900 python3 -m pytest $TESTS -x -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }
  tail -3 gpurun_out/${TAG}_tests.log
fi
AB_ARGS="--shape 1" bash scripts/ab_variants.sh gpurun_out/${TAG}_ab_wide.txt "$@" > /dev/null
AB_ARGS="--shape 0" bash scripts/ab_variants.sh gpurun_out/${TAG}_ab_narrow.txt "$@" > /dev/null
echo "== wide"; cut -c1-100 gpurun_out/${TAG}_ab_wide.txt
echo "== narrow"; cut -c1-100 gpurun_out/${TAG}_ab_narrow.txt
