#!/bin/bash
set -u
OUT=gpurun_out/r05l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_hip_gemm.py tests/test_hip_production_dispatch.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
timeout 600 python3 tools/tn_gather_cost.py > $OUT/r05_tn_gather_cost_after.txt 2>&1
grep -v amdgpu.ids $OUT/r05_tn_gather_cost_after.txt
bash tools/ab_trees.sh .ab_old/r04 3 2>&1 | tee $OUT/r05_ab_vs_round4.txt
STSWIN_SHAPE_PROFILE=1 timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --profile-stride 1 --dump-prof $OUT/r05_gemm_shapes_in_step.txt > $OUT/bench_shapes.log 2>&1
grep gemm_tn $OUT/r05_gemm_shapes_in_step.txt | head -24
