mkdir -p gpurun_out/r03i
run() { echo "== $1"; OATK_HIP_LIB=$PWD/oatk_amd/lib/$1 python tools/kbench.py --reads 400000 --steps 4 2>&1 | tail -2 | head -1 | cut -c1-80; }
( for rep in 1 2; do run liboatk_hip.so; run var_e1.so; run var_e5.so; done
) >> gpurun_out/r03i/b_phase_experiments.txt 2>&1
tail -12 gpurun_out/r03i/b_phase_experiments.txt
