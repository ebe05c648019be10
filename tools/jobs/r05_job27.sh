#!/bin/bash
# round 5, GPU job 27: torch (aten) launches left in the contrastive step
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/torch_ops.py contrast 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" | tee $OUT/r05_torch_ops_contrast.txt | head -60
