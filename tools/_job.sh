mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_hip_conv.py -x -q -k "split_tile or second_output" 2>&1 | grep -v amdgpu.ids | tail -12
for i in 1 2; do for f in 6 19; do
REFID_DOWN_SPLIT=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r06/b8_down$f.$i.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r06/b8_down$f.$i.json').read().strip().splitlines()[-1]); print('down', $f, d['ms_per_step'], d['value'])
"
done; done
