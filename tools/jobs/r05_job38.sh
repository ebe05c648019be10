#!/bin/bash
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
STSWIN_HIP_LIB=$PWD/.ab_old/tuning/libstswin_hip.so python3 tools/ragged_rounds.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_ragged_rounds.txt
