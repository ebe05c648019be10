#!/bin/bash
# round 5, GPU job 36: four weight gradients of a Swin block in one launch (fc2 joins; LayerNorm backward into a new buffer) - tests, step A/B
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_hip_rowops.py tests/test_hip_gemm.py -m gpu -x -q -k "layernorm or group" 2>&1 | tail -2
timeout 900 python3 -m pytest tests/test_hip_swin.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
  for v in off three four; do
    unset STSWIN_TN_GROUP STSWIN_NO_TN_GROUP4
    if [ $v = off ]; then export STSWIN_TN_GROUP=0; fi
    if [ $v = three ]; then export STSWIN_NO_TN_GROUP4=1; fi
    python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('grouped weight gradients: $v', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
  done
done 2>&1 | tee $OUT/r05_tn_group4_in_step_ab.txt
unset STSWIN_TN_GROUP STSWIN_NO_TN_GROUP4
STSWIN_SHAPE_PROFILE=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --profile-stride 1 --dump-prof $OUT/r05_gemm_shapes_in_step_grouped.txt > /dev/null 2>&1
grep "gemm_tn" $OUT/r05_gemm_shapes_in_step_grouped.txt | head -12
