#!/bin/bash
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
export STSWIN_HIP_LIB=$PWD/.ab_old/tuning/libstswin_hip.so
{ python3 tools/tn_group_timeline.py; echo "== all three plain"; PLAIN=1 python3 tools/tn_group_timeline.py; echo "== weight 100"; STSWIN_TN_GROUP_W=100 python3 tools/tn_group_timeline.py; } 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_tn_group_timeline.txt
