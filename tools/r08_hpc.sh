#!/bin/bash
# kernel A after a change: parity of the scan and the reader, then its time at 400 k reads (three runs) and inside a config-3 step (two runs)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
MC="python tools/memcap.py --rss-gb 300 --timeout"
$MC 900 -- python -m pytest tests/test_gpu_scan.py tests/test_gpu_ingest.py tests/test_gpu_append.py tests/test_gpu_errors.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2 3; do $MC 600 -- python tools/kbench.py --reads 400000 --steps 5 2>&1 | grep -E "^reads" | cut -c1-75; done
for i in 1 2; do
  $MC 600 -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = b['phases_ms']
print(b['ms_per_step'], 'hpc', p['hpc'], 'syncmer', p['syncmer'], 'kmer_hash', p['kmer_hash'], 'count_group', p['count_group'], 'ec_solve', p['ec_solve'])"
done
