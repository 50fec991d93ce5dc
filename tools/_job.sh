set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_hip_conv.py -x -q -k "wino" > gpurun_out/r06/t_wino.txt 2>&1
tail -30 gpurun_out/r06/t_wino.txt
timeout 600 python tools/bench_wino6.py > gpurun_out/r06/bench_wino6_f16.txt 2>&1
cat gpurun_out/r06/bench_wino6_f16.txt | grep -v amdgpu.ids
