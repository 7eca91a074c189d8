#!/bin/bash
# The round's evidence on the final code, every step under the memory watchdog:  gpurun --timeout 5000 -- 'bash tools/r07_final.sh r07z'
#   bench.py (the driver's command), rocprofv3 kernel statistics of a config-3 step, the PMC passes (HBM bytes, kernel B's instruction count, the clock),
#   the config-1 surrogate's solver under rocprofv3 (statistics + PMC), the surrogate's CLI under rocprofv3.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
TAG=${1:-r07z}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
MC="python tools/memcap.py --rss-gb 400 --timeout"
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -12; nproc; free -g | head -2; /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | head -20) > $O/${TAG}_env.txt 2>&1
SECONDS=0
$MC 2400 -- python bench.py > $O/${TAG}_bench.json 2> $O/bench.err; echo "bench rc=$? (${SECONDS}s)"; tail -c 600 $O/${TAG}_bench.json
( cd /tmp && python $GRAFT_REPO_ROOT/tools/memcap.py --rss-gb 400 --timeout 900 -- rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/${TAG}_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof.err ); echo "rocprof rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${TAG}_kernel_stats.csv && cut -c1-140 $f | head -14
bash tools/pmc_r02.sh $TAG > $O/pmc.log 2>&1; echo "pmc rc=$?"; tail -30 $O/pmc.log | cut -c1-200
bash tools/prof_config1s.sh $TAG 200000 > $O/c1s.log 2>&1; echo "config1s prof rc=$?"; head -14 $O/${TAG}_config1s_kernel_stats.csv | cut -c1-140
bash tools/prof_cli_config1s.sh $TAG > $O/cli_prof.log 2>&1; echo "cli prof rc=$?"; grep "oatk_dropin\] [a-z_]* *[0-9]" $O/cli.log | head -8
ls $O
