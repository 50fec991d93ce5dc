timeout 1200 python -m pytest tests/test_hip_ddp.py -x -q -s -k "graph_replay_with" 2>&1 | grep -v amdgpu.ids | tail -8
