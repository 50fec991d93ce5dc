mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r06/t_all2.txt 2>&1
tail -8 gpurun_out/r06/t_all2.txt
