#!/bin/bash
# round 5, GPU job 4: the overlap proxy (verdict item 5) on the product build, then the corrected in-kernel tile timeline (tuning build)
set -u
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python3 tools/overlap_proxy.py > $OUT/r05_overlap_proxy.txt 2>&1
cat $OUT/r05_overlap_proxy.txt
STSWIN_TUNING=1 timeout 900 python3 __graft_entry__.py --force > $OUT/build_tuning.log 2>&1 || tail -20 $OUT/build_tuning.log
for epi in plain gelu gelu_dgelu mul_r resid; do
  STSWIN_TL_EPI=$epi timeout 300 python3 tools/gemm_timeline.py 65536 2048 512 >> $OUT/r05_gemm_tile_timeline.txt 2>&1
done
STSWIN_TL_EPI=gelu_dgelu timeout 300 python3 tools/gemm_timeline.py 16384 4096 1024 >> $OUT/r05_gemm_tile_timeline.txt 2>&1
STSWIN_TL_EPI=gelu_dgelu STSWIN_TL_VARIANT=$((128 + 16777216)) STSWIN_TL_TILE=128,256 timeout 300 python3 tools/gemm_timeline.py 65536 2048 512 >> $OUT/r05_gemm_tile_timeline.txt 2>&1
STSWIN_TL_EPI=plain STSWIN_TL_VARIANT=$((128 + 16777216)) STSWIN_TL_TILE=128,256 timeout 300 python3 tools/gemm_timeline.py 65536 2048 512 >> $OUT/r05_gemm_tile_timeline.txt 2>&1
grep -v amdgpu.ids $OUT/r05_gemm_tile_timeline.txt
timeout 600 python3 -m pytest tests/test_hip_gemm.py -m gpu -x -q -k "duo or ring_register" 2>&1 | tail -3
