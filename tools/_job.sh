timeout 1800 python -m pytest tests/test_hip_network.py -q -x 2>&1 | tail -15
