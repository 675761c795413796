#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
PMC_PROBE_ARGS="--shape 1 --no-ellipsoid" bash scripts/run_pmc_sets.sh r05h \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
  "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" \
  "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC_BANDWIDTH SQ_INSTS_LDS_LOAD_BANDWIDTH SQ_INSTS_LDS_STORE_BANDWIDTH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
  -- base abl1 hot4 hot1f > /dev/null 2>&1
python3 - <<'PY'
import collections
cur=None; d=collections.OrderedDict()
for l in open('gpurun_out/pmcsets_r05h.txt'):
    if l.startswith('## '): cur=l[3:].strip(); d.setdefault(cur,{}); continue
    if 'scan_wide_kernel' in l:
        parts=l.split()
        for i,t in enumerate(parts):
            if t.startswith('SQ'):
                d[cur][t]=float(parts[i+1]); break
names=list(d.keys())
ctrs=sorted({c for v in d.values() for c in v})
print("%-32s"%"counter (filled, wide)"+"".join("%12s"%n for n in names))
for c in ctrs:
    print("%-32s"%c+"".join("%12.4g"%d[n].get(c,float('nan')) for n in names))
PY
