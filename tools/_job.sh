mkdir -p gpurun_out/r06
WINO6_TERMS=3 WINO6_ONLY=20,21 WINO6_TAG=h python tools/probes/wino6_ablate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/wino6_abl_f16_dispatch.txt
cat gpurun_out/r06/wino6_abl_f16_dispatch.txt
