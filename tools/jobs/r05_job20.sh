#!/bin/bash
# round 5, GPU job 20: BatchNorm kernels with cold operands (rotated over > 1 GB) against the warm numbers of tools/bench_bn.py so far
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
{ python3 tools/bench_bn.py; STSWIN_BN_COLD=0 python3 tools/bench_bn.py; } 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_batchnorm_cold_vs_warm.txt
