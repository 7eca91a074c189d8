cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r06e; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
OATK_DEBUG_EC_STAGES=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python $GRAFT_REPO_ROOT/tools/solverbench.py --workload ${1:-config1s} --reads ${2:-200000} --reps 3 > $O/log.txt 2>&1
grep -v "^[WEI]2026" $O/log.txt | grep -v "ec stages" | tail -3; grep "second stage" $O/log.txt
python - <<PY
import csv,glob
f=glob.glob("$O/tr/**/t_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if any(x in r["Kernel_Name"] for x in ("ec_fused", "ec_wave_kernel", "ec_heavy", "ec_route", "ec_list", "ec_count_blocks", "ec_live", "ec_assemble", "ec_new_n", "radix"))]
t0=min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    print("%-40s start %9.3f ms dur %9.3f ms grid %s wg %s lds %s" % (r["Kernel_Name"].split("(")[0][-40:], (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("Grid_Size_X"), r.get("Workgroup_Size_X"), r.get("LDS_Block_Size")))
PY
