mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r06/t_all.txt 2>&1
tail -15 gpurun_out/r06/t_all.txt
