#!/bin/bash
# Cross-compile everything tools/profile_round.sh runs on the GPU box (CPU container, before `gpurun`):
#   tools/probes/bin/librefid_w6abl*.so   conv_wino6.hip with one piece removed   (tools/probes/wino6_ablate.py)
#   tools/probes/bin/librefid_wwabl*.so   wgrad_wino.hip with one piece removed   (tools/probes/wgrad_wino_ablate.py)
#   tools/probes/bin/librefid_w24abl*.so  wgrad_wino24.hip with one piece removed (tools/probes/w24_ablate.py)
#   tools/probes/bin/{wino6_loop,mfma_lds_feed,mfma_valu_overlap}                 stand-alone hardware probes
# The variant libraries link the CURRENT objects of the product build, so they must be rebuilt whenever csrc/ changes
# (round 3's r03_wino6_ablation.txt was a Python traceback for exactly that reason); the ablation scripts refuse to run
# against stale libraries, and this script ends by restoring the product build of refid_amd/librefid_hip.so.
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python tools/probes/wino6_ablate.py --build
python tools/probes/wgrad_wino_ablate.py --build
python tools/probes/w24_ablate.py --build
python -m refid_amd.build > /dev/null                      # back to the product library
mkdir -p tools/probes/bin
for p in wino6_loop mfma_lds_feed mfma_valu_overlap; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/probes/bin/$p tools/probes/$p.hip
done
python - <<'PY'
from refid_amd import _lib
L = _lib.lib()
assert L.refid_experimental_tiles() == 0, "product library expected after build_probes.sh"
print("probes built; product library restored (ABI", L.refid_abi_version(), ")")
PY
