#!/bin/bash
# round 5, GPU job 29: two late weight gradients of a Swin block as one resident set (two streams) against two launches
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
{ python3 tools/tn_pair_concurrent.py; MK=16384 CC=1024 python3 tools/tn_pair_concurrent.py; } 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_tn_pair_concurrent.txt
