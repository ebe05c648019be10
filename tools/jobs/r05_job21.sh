#!/bin/bash
# round 5, GPU job 21: BatchNorm backward with the channel constants handed out through LDS - tests, then workgroup-count sweep, cold operands
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_hip_headops.py tests/test_hip_resnet.py -m gpu -x -q 2>&1 | tail -2
for w in 512 1024 2048 4096; do
  echo "== reduce and dx target $w workgroups"
  STSWIN_BN_RED_WGS=$w STSWIN_BN_DX_WGS=$w python3 tools/bench_bn.py 2>&1 | grep -v "amdgpu.ids\|^#"
done | tee $OUT/r05_bn_bwd_lds_constants_sweep.txt
