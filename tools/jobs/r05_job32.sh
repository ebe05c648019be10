#!/bin/bash
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3; do
  for v in off 100 120 140; do
    if [ $v = off ]; then export STSWIN_TN_GROUP=0; unset STSWIN_TN_GROUP_W; else export STSWIN_TN_GROUP=1; export STSWIN_TN_GROUP_W=$v; fi
    python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('grouped weight gradients $v', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms', 'gemm_tn', round(d['roofline']['other_kernels']['gemm_tn_bf16']['ms_per_step'],3), 'ms')"
  done
done 2>&1 | tee $OUT/r05_tn_group_in_step_ab.txt
