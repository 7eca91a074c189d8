#!/bin/bash
# The GPU tests one FILE at a time, each under its own timeout, with the device's and the host's memory noted before each file -- for finding which file a box does not
# survive (round 5 lost three boxes to the suite as a whole without learning which test it was).  Logs under gpurun_out/suite/ as it goes.
#   gpurun --timeout 3000 -- 'bash tools/gpu_suite_by_file.sh [first-file-pattern]'
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out/suite
start="${1:-}"
for f in tests/test_gpu_*.py; do
    if [ -n "$start" ] && [[ "$f" < "tests/test_gpu_$start" ]]; then continue; fi
    {
        echo "== $f  $(date +%T)"
        rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -1
        free -g | awk 'NR==2{print "host used GB", $3}'
    } >> gpurun_out/suite/progress.txt
    python tools/memcap.py --rss-gb 400 --timeout 900 -- python -m pytest "$f" -x -q -m gpu > "gpurun_out/suite/$(basename "$f" .py).txt" 2>&1
    echo "   rc $?  $(tail -n 1 "gpurun_out/suite/$(basename "$f" .py).txt")" >> gpurun_out/suite/progress.txt
done
cat gpurun_out/suite/progress.txt
