timeout 1200 python -m pytest tests/test_hip_configs.py -x -q -m gpu -k "two_ranks" 2>&1 | tail -15
