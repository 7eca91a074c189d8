#!/bin/bash
# Round 6, first GPU call: the code of round 5 that compiled and never ran (ec_rows.hpp, ec_fused.hpp CERT, multi_host.c's thread-start gate), each under the
# memory watchdog (tools/memcap.py) and its own timeout, most informative first; then the surrogate's solver with and without the table test.
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/${1:-r06a}; mkdir -p $O
export TMPDIR=/tmp
MC="python tools/memcap.py --rss-gb 200 --timeout"
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -6; nproc; free -g | head -2) > $O/env.txt 2>&1
export OATK_TEST_EC_ROWS=1 OATK_TEST_THREAD_FAIL=1
$MC 900 -- python -m pytest tests/test_gpu_levdist.py -x -q -m gpu > $O/levdist.txt 2>&1; echo "levdist rc $? $(tail -n 1 $O/levdist.txt)"
$MC 900 -- python -m pytest tests/test_gpu_ec.py -q -m gpu > $O/ec.txt 2>&1; echo "ec rc $? $(tail -n 1 $O/ec.txt)"
$MC 600 -- python -m pytest tests/test_gpu_cli.py -q -m gpu -k "thread" > $O/cli_thread.txt 2>&1; echo "cli-thread rc $? $(tail -n 1 $O/cli_thread.txt)"
unset OATK_TEST_EC_ROWS OATK_TEST_THREAD_FAIL
$MC 300 -- python tools/solverbench.py --workload config1s --reads 200000 --reps 2 --set "OATK_DEBUG_EC_CERT=0" --set "OATK_DEBUG_EC_CERT=1" > $O/solver_c1s.txt 2>&1; echo "solverbench rc $?"; tail -4 $O/solver_c1s.txt
EC_EFFORT_WARM=1 $MC 300 -- python tools/ec_effort.py config1s 200000 > $O/effort_nocert.txt 2>&1; echo "effort rc $?"; grep -E "ec_solve|ec [0-9]|stages|longest" $O/effort_nocert.txt | head
OATK_DEBUG_EC_CERT=1 EC_EFFORT_WARM=1 $MC 300 -- python tools/ec_effort.py config1s 200000 > $O/effort_cert.txt 2>&1; echo "effort cert rc $?"; grep -E "ec_solve|ec [0-9]|stages|longest" $O/effort_cert.txt | head
