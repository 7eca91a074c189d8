#!/bin/bash
# liboatk_hip_experiments.so: the library's sources with -DOATK_EXPERIMENTS (Myers' bit-vector edit distance and its A/B entry point, tools/experiments/oatk_experiments.h).
#   bash tools/experiments/build.sh && OATK_HIP_LIB=tools/experiments/liboatk_hip_experiments.so python tools/edbench.py
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DOATK_EXPERIMENTS -o $R/tools/experiments/liboatk_hip_experiments.so $R/oatk_amd/csrc/*.hip
echo built $R/tools/experiments/liboatk_hip_experiments.so
