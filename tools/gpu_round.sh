#!/bin/bash
# One GPU session: parity tests, smoke, bench, and a rocprofv3 kernel trace of the same bench command.
# Usage (from the repo root, through gpurun):  bash tools/gpu_round.sh <tag> [pytest-args]
set -u
TAG=${1:-r02}
shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo ==" > $OUT/env.txt
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -12; nproc; free -g | head -2) >> $OUT/env.txt 2>&1
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q "$@" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? (${SECONDS}s)" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/smoke.log
SECONDS=0
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? (${SECONDS}s)"
tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$OUT/bench_prof.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err ); echo "rocprof rc=$?"
find $OUT/prof -name "*stats*" | head
