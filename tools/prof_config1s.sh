#!/bin/bash
# The config-1 surrogate's EC round under rocprofv3: kernel stats, then the PMC counters of the solver's kernels, each group in a pass of its own (counters only with
# --kernel-trace: gpurun refuses anything else beside --pmc).  VERDICT r04 item 8 asks for both under profiles/; round 5 ran out of GPU boxes before taking them.
#   gpurun --timeout 1500 -- 'bash tools/prof_config1s.sh r06a [reads]'     then copy gpurun_out/<tag>/*_stats.csv and *_pmc_ec.csv into profiles/
TAG=${1:-c1s}; N=${2:-200000}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/memcap.py --rss-gb 200 --timeout 500 -- python $R/tools/solverbench.py --workload config1s --reads $N --reps 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o t -- $CMD > $O/stats.log 2>&1; echo "stats rc=$?"
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_config1s_kernel_stats.csv && head -12 $f
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/c1s_a -o p -- $CMD > $O/c1s_a.log 2>&1; echo "pmc a rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $O/c1s_b -o p -- $CMD > $O/c1s_b.log 2>&1; echo "pmc b rc=$?"
python - <<PY
import csv, collections, glob
out = open("$O/${TAG}_config1s_pmc_ec.csv", "w")
out.write('kernel,counter,"sum over dispatches (tools/prof_config1s.sh: tools/solverbench.py --workload config1s --reads $N --reps 1; rocprofv3 --kernel-trace --pmc, two passes)"\n')
for d in ("c1s_a", "c1s_b"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob("$O/%s/**/p_counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if not any(x in k for x in ("ec_wave", "ec_heavy", "ec_fused", "ec_slab", "ec_route")): continue
            agg[k.split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        for a, b in sorted(v.items()):
            out.write('"%s",%s,%d\n' % (k, a, b))
out.close()
print(open("$O/${TAG}_config1s_pmc_ec.csv").read()[:4000])
print(open("$O/stats.log").read()[-800:])
PY
