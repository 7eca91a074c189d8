#!/bin/bash
# builds and runs the two micro-benchmarks on the GPU box; output goes to gpurun_out/ubench/ (copy what is cited into profiles/)
set -e
D=$(dirname "$0"); O=$GRAFT_REPO_ROOT/gpurun_out/ubench; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o $D/valu_rates $D/valu_rates.hip
$D/valu_rates | tee $O/valu_rates.txt
