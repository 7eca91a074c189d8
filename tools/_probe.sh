mkdir -p gpurun_out/r03l
run() { echo "== $1"; OATK_HIP_LIB=$PWD/oatk_amd/lib/$1 python tools/kbench.py --reads 400000 --steps 4 2>&1 | tail -2 | head -1 | cut -c1-60; }
( for rep in 1 2; do run liboatk_hip.so; run var_a1.so; run var_a2.so; run var_a3.so; done
) > gpurun_out/r03l/a_phase_experiments.txt 2>&1
cat gpurun_out/r03l/a_phase_experiments.txt
