#!/bin/bash
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
for w in 100 110 120 130 150 180; do echo "== weight of a problem with a row map: $w %"; STSWIN_TN_GROUP_W=$w ONLY_STEP=1 python3 tools/tn_group_probe.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/r05_tn_group_weight_sweep.txt
