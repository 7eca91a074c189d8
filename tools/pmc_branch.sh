#!/bin/bash
# branch and instruction-fetch counters of the error-block solver (counters only with --kernel-trace): bash tools/pmc_branch.sh   (through gpurun)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcbr; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | grep -i "BRANCH\|IFETCH\|INST_CYCLES\|INSTS_\|WAIT_IFETCH\|ICACHE\|SQC" | tr '\n' ' ' | head -c 3000; echo
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -o p -- python $R/tools/solverbench.py --workload config2 --reps 1 > $O/a.log 2>&1
tail -3 $O/a.log
python - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
try:
    for r in csv.DictReader(open("$O/a/p_counter_collection.csv")):
        k = r["Kernel_Name"]
        if "ec_wave" not in k: continue
        agg[k.split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items(): print(k, dict(v))
except Exception as e: print("no csv", e)
PY
