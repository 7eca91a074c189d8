mkdir -p gpurun_out/r03j
run() { echo "== $1"; OATK_HIP_LIB=$PWD/oatk_amd/lib/$1 python tools/kbench.py --reads 400000 --steps 4 2>&1 | tail -2 | head -1 | cut -c1-80; }
( timeout 900 python -m pytest tests/test_gpu_scan.py -m gpu -q -x 2>&1 | tail -5
for rep in 1 2; do run var_base.so; run var_prev.so; run liboatk_hip.so; done
python tools/tiebench.py
) > gpurun_out/r03j/b_inlane_decision.txt 2>&1
cat gpurun_out/r03j/b_inlane_decision.txt
