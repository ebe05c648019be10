#!/bin/bash
set -u
OUT=gpurun_out/r05k
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python3 tools/tn_gather_cost.py > $OUT/r05_tn_gather_cost.txt 2>&1
grep -v amdgpu.ids $OUT/r05_tn_gather_cost.txt
timeout 1800 python3 -m pytest tests/test_hip_configs.py -m gpu -x -q -s -k "config4_full_size" > $OUT/pytest_new.log 2>&1
grep -v "Warning\|warn" $OUT/pytest_new.log | grep "configs\[4\] fp8-step\|passed\|failed\|Error" | head
