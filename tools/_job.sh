set -x
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_hip_network.py tests/test_hip_train_step.py -q > gpurun_out/r06/t_net.txt 2>&1
tail -25 gpurun_out/r06/t_net.txt
