#!/bin/bash
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
export STSWIN_HIP_LIB=$PWD/.ab_old/tuning/libstswin_hip.so
{ python3 tools/gemm_timeline.py tn; for sp in 8 10 16; do echo "== forced $sp splits"; STSWIN_TL_SPLITS=$sp python3 tools/gemm_timeline.py tn; done; } 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_tn_stage_times.txt
