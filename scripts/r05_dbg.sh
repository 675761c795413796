#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
for v in ${DBG_VARIANTS:-nomask m1 m2 m4 base}; do
  if [ "$v" = base ]; then unset TISSUE_SCAN_LIB; else export TISSUE_SCAN_LIB=$R/scratch/lib$v.so; fi
  echo "== $v" >> gpurun_out/${DBG_OUT:-r05_dbg}.txt
  timeout -k 10 300 python3 -m pytest tests/test_gpu_sweep_parity.py -q -m gpu -k "sweep" 2>&1 | tail -8 | cut -c1-200 >> gpurun_out/${DBG_OUT:-r05_dbg}.txt
done
cat gpurun_out/${DBG_OUT:-r05_dbg}.txt
