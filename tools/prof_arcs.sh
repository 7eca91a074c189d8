#!/bin/bash
# builds the -DECF_PROF2 development library (CPU, ~2 min):  bash tools/prof_arcs.sh ; then on a GPU box: OATK_HIP_LIB=tools/experiments/liboatk_hip_prof2.so python tools/prof_arcs.py
cd "$(dirname "$0")/.." && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DECF_PROF2 -o tools/experiments/liboatk_hip_prof2.so oatk_amd/csrc/api.hip
