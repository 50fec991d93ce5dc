timeout 600 python -m pytest tests/test_hip_conv.py -x -q -k "wino" 2>&1 | tail -3
timeout 600 python tools/bench_wino6.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_khmajor.txt; cat gpurun_out/r06_khmajor.txt | cut -c1-150
