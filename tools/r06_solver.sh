#!/bin/bash
# Round 6: the solver's tests and the surrogate's solver A/B, under the memory watchdog.   gpurun -- 'bash tools/r06_solver.sh <tag> [settings ...]'
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/${1:-r06s}; shift; mkdir -p $O
export TMPDIR=/tmp
MC="python tools/memcap.py --rss-gb 200 --timeout"
$MC 900 -- python -m pytest tests/test_gpu_ec.py -q -m gpu -x > $O/ec.txt 2>&1; echo "ec rc $? $(tail -n 2 $O/ec.txt | head -1)"
args=(); for s in "$@"; do args+=(--set "$s"); done
OATK_DEBUG_EC_STAGES=1 $MC 600 -- python tools/solverbench.py --workload config1s --reads 200000 --reps 2 "${args[@]}" > $O/solver_c1s.txt 2>&1; echo "solverbench rc $?"; grep -v "^\[ec stages\]" $O/solver_c1s.txt | tail -8; grep "ec stages" $O/solver_c1s.txt | tail -7
EC_EFFORT_WARM=1 $MC 300 -- python tools/ec_effort.py config1s 200000 > $O/effort.txt 2>&1; echo "effort rc $?"; grep -E "^config1s|time on|waves:" $O/effort.txt; grep -A4 "heaviest by time" $O/effort.txt
$MC 300 -- python tools/solverbench.py --workload config3 --reps 3 > $O/solver_c3.txt 2>&1; tail -2 $O/solver_c3.txt
