#!/bin/bash
# PMC passes (counters only with --kernel-trace, as the pool requires) over tools/kbench.py; usage: bash tools/pmc.sh <tag> [reads]
TAG=${1:-pmc}; READS=${2:-20000}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/a -o p -- python $R/tools/kbench.py --reads $READS --steps 1 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/b -o p -- python $R/tools/kbench.py --reads $READS --steps 1 > $O/b.log 2>&1
python - <<PY
import csv, collections
for d in 'ab':
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open('$O/%s/p_counter_collection.csv' % d)):
        k=r['Kernel_Name']
        if 'oatk::hpc' not in k and 'oatk::syncmer' not in k: continue
        agg[k[:48]][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        print(k, {a: '%.3g'%b for a,b in sorted(v.items())})
PY
