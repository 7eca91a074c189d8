#!/bin/bash
# PMC counters of the error-block solver (counters only with --kernel-trace, each group in its own pass): bash tools/pmc_ec.sh <tag> [workload]
TAG=${1:-ec}; WL=${2:-config2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/a -o p -- python $R/tools/solverbench.py --workload $WL --reps 1 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $O/b -o p -- python $R/tools/solverbench.py --workload $WL --reps 1 > $O/b.log 2>&1
python - <<PY
import csv, collections
out = open("$O/${TAG}_pmc_ec.csv", "w")
out.write('kernel,counter,"sum over dispatches (tools/pmc_ec.sh: tools/solverbench.py --workload $WL --reps 1; rocprofv3 --kernel-trace --pmc, two passes)"\n')
for d in "ab":
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open("$O/%s/p_counter_collection.csv" % d)):
        k = r["Kernel_Name"]
        if "ec_wave" not in k and "ec_quad" not in k: continue
        agg[k.split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        for a, b in sorted(v.items()):
            out.write('"%s",%s,%d\n' % (k, a, b))
out.close()
print(open("$O/${TAG}_pmc_ec.csv").read())
print(open("$O/a.log").read()[-600:])
PY
