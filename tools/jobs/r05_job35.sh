#!/bin/bash
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_hip_gemm.py -m gpu -x -q -k "group" 2>&1 | tail -2
{ for w in 100 105 110 120; do echo "== static-slot kernel, weight $w"; STSWIN_HIP_LIB=$PWD/.ab_old/tuning/libstswin_hip.so STSWIN_TN_GROUP_W=$w python3 tools/tn_group_timeline.py 2>&1 | grep -v amdgpu.ids | head -4; done; } | tee $OUT/r05_tn_group_timeline_static.txt
for i in 1 2 3; do
  for v in off 100 105 110 120; do
    if [ $v = off ]; then export STSWIN_TN_GROUP=0; unset STSWIN_TN_GROUP_W; else export STSWIN_TN_GROUP=1; export STSWIN_TN_GROUP_W=$v; fi
    python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('grouped weight gradients $v', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
  done
done 2>&1 | tee $OUT/r05_tn_group_static_in_step_ab.txt
