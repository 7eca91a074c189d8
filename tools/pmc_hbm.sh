#!/bin/bash
# HBM traffic of the scan kernels from the L2 memory-side counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in
# their own passes, counters only with --kernel-trace.  usage: bash tools/pmc_hbm.sh <tag>
TAG=${1:-hbm}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-syncerr > $O/$c.log 2>&1
done
python - <<PY
import csv, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$O/%s/p_counter_collection.csv" % c)):
        k = r["Kernel_Name"]
        if "oatk::" in k: agg[k[:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(c, k, "dispatches", len(v), "mean per dispatch", sum(v) / len(v))
PY
