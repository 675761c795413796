#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/r05_counters_list.txt 2>&1
cd $R
grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_LEVEL_WAVES\|SQ_BUSY_CU_CYCLES\|SQ_INSTS_[A-Z_]*\|SQ_WAVE_[A-Z_]*" gpurun_out/r05_counters_list.txt | sort -u | tr '\n' ' ' > gpurun_out/r05_counter_names.txt
cat gpurun_out/r05_counter_names.txt
