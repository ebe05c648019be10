#!/bin/bash
# round 5, GPU job 30: grouped weight-gradient launch (fc1 + proj + qkv of a Swin block) - tests, step A/B, per-shape table
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_hip_gemm.py -m gpu -x -q -k "group" 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_hip_swin.py tests/test_hip_bf16_stages.py tests/test_hip_model.py tests/test_hip_configs.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
  for v in 0 1; do
    STSWIN_TN_GROUP=$v python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('grouped weight gradients $v', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms', 'gemm_tn', round(d['roofline']['other_kernels']['gemm_tn_bf16']['ms_per_step'],3), 'ms', round(d['roofline']['other_kernels']['gemm_tn_bf16']['tflops'],1), 'TF/s')"
  done
done 2>&1 | tee $OUT/r05_tn_group_in_step_ab.txt
STSWIN_SHAPE_PROFILE=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --profile-stride 1 --dump-prof $OUT/r05_gemm_shapes_in_step_grouped.txt > /dev/null 2>&1
grep "gemm_tn" $OUT/r05_gemm_shapes_in_step_grouped.txt | head -30
