#!/bin/bash
# PMC evidence for a round (counters only with --kernel-trace, each group in its own pass, as the pool requires):
#   <tag>_pmc_scan.csv          SQ counters of the two scan kernels over tools/kbench.py (20 000 reads: 2 dispatches each)
#   <tag>_pmc_hbm_config3.csv   FETCH_SIZE / WRITE_SIZE per dispatch of every oatk kernel in one bench.py step at config 3 (the headline workload)
#   <tag>_pmc_clock_config3.csv GRBM_GUI_ACTIVE over the duration of the same dispatches: the clock the kernels ran at (bench.py prices the VALU ceiling with it)
# usage (through gpurun): bash tools/pmc_r02.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/a -o p -- python $R/tools/kbench.py --reads 20000 --steps 1 > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/b -o p -- python $R/tools/kbench.py --reads 20000 --steps 1 > $O/b.log 2>&1
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/$c.log 2>&1
done
python - <<PY
import csv, collections
import re
pos = int(re.search(r"hoco positions per dispatch (\d+)", open("$O/a.log").read()).group(1))
out = open("$O/${TAG}_pmc_scan.csv", "w")
out.write('kernel,counter,"sum_over_dispatches (tools/pmc_r02.sh: tools/kbench.py --reads 20000 --steps 1 = 2 dispatches per kernel, positions=%d hoco positions in all; rocprofv3 --kernel-trace --pmc, two passes)"\n' % (2 * pos))
for d in "ab":
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open("$O/%s/p_counter_collection.csv" % d)):
        k = r["Kernel_Name"]
        if "oatk::hpc" not in k and "oatk::syncmer" not in k: continue
        agg[k.split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(agg.items()):
        for a, b in sorted(v.items()):
            out.write('"%s",%s,%d\n' % (k, a, b))
out.close()
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$O/%s/p_counter_collection.csv" % c)):
        k = r["Kernel_Name"]
        if "oatk::" in k or "ec_" in k: agg[k].append((int(r["Grid_Size"]) if "Grid_Size" in r else 0, float(r["Counter_Value"])))
    for k, v in agg.items():
        g = max(x[0] for x in v)
        big = [x[1] for x in v if x[0] == g]
        per.setdefault(k, {})[c] = (sum(big) / len(big), len(big))
out = open("$O/${TAG}_pmc_hbm_config3.csv", "w")
out.write('kernel,FETCH_SIZE_KiB_per_dispatch,WRITE_SIZE_KiB_per_dispatch,"note: rocprofv3 --kernel-trace --pmc, one counter per pass (tools/pmc_r02.sh), bench.py config3 (2000000 reads), mean over the dispatches with the largest grid of each kernel; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide streaming reads, so bench.py doubles it (MI355X_MICROARCH.md, HBM section)"\n')
for k, v in sorted(per.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", (0, 0))[0] + kv[1].get("WRITE_SIZE", (0, 0))[0])):
    out.write('"%s",%.1f,%.1f\n' % (k, v.get("FETCH_SIZE", (0, 0))[0], v.get("WRITE_SIZE", (0, 0))[0]))
out.close()
# the clock: GRBM_GUI_ACTIVE (cycles the graphics engine was busy) of a dispatch over its duration in the same pass's kernel trace
try:
    dur = {}
    for r in csv.DictReader(open("$O/GRBM_GUI_ACTIVE/p_kernel_trace.csv")):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$O/GRBM_GUI_ACTIVE/p_counter_collection.csv")):
        k = r["Kernel_Name"]
        if ("oatk::" in k or "ec_" in k) and r["Dispatch_Id"] in dur:
            agg[k].append((int(r["Grid_Size"]) if "Grid_Size" in r else 0, float(r["Counter_Value"]), dur[r["Dispatch_Id"]]))
    out = open("$O/${TAG}_pmc_clock_config3.csv", "w")
    out.write('kernel,GRBM_GUI_ACTIVE_per_dispatch,duration_ns_per_dispatch,GHz,"note: tools/pmc_r02.sh, bench.py config3, the dispatches with the largest grid of each kernel; kernels of >= 1 ms only (the counter is per dispatch, its window a little wider than the kernel); rocprofv3 reports the SUM over the eight XCDs of the part (one GRBM each), so GHz = counter / 8 / duration"\n')
    for k, v in sorted(agg.items(), key=lambda kv: -max(x[2] for x in kv[1])):
        g = max(x[0] for x in v)
        big = [x for x in v if x[0] == g]
        cyc, ns = sum(x[1] for x in big) / len(big), sum(x[2] for x in big) / len(big)
        if ns >= 1e6: out.write('"%s",%.0f,%.0f,%.3f\n' % (k, cyc, ns, cyc / 8 / ns))
    out.close()
    print(open("$O/${TAG}_pmc_clock_config3.csv").read()[:2000])
except Exception as ex:
    print("clock pass:", ex)
print(open("$O/${TAG}_pmc_hbm_config3.csv").read()[:3000])
print(open("$O/${TAG}_pmc_scan.csv").read())
PY
