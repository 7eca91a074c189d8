#!/bin/bash
# runs the micro-benchmarks on the GPU box (built here by hipcc; the binaries travel with the snapshot); output goes to gpurun_out/ubench/
# (copy what is cited into profiles/)
D=$(dirname "$0"); O=$GRAFT_REPO_ROOT/gpurun_out/ubench; mkdir -p $O
[ -x $D/valu_rates ] || hipcc --offload-arch=gfx950 -O3 -o $D/valu_rates $D/valu_rates.hip
[ -x $D/pin_rates ] || hipcc --offload-arch=gfx950 -O2 -o $D/pin_rates $D/pin_rates.hip -lpthread
(nproc; free -g | head -2; cat /sys/kernel/mm/transparent_hugepage/enabled) > $O/host.txt 2>&1
timeout 600 $D/valu_rates > $O/valu_rates.txt 2>&1; echo "valu_rates rc=$?"
timeout 600 $D/pin_rates 8 32 > $O/pin_rates.txt 2>&1; echo "pin_rates rc=$?"
