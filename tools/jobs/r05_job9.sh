#!/bin/bash
# round 5, GPU job 9: in-step per-shape GEMM tables with and without the start-time stagger; remaining new tests
set -u
OUT=gpurun_out/r05i
mkdir -p $OUT
export TMPDIR=/tmp
for st in 0 1; do
  STSWIN_NT_STAGGER=$st STSWIN_SHAPE_PROFILE=1 timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --profile-stride 1 --dump-prof $OUT/r05_gemm_shapes_in_step_stagger$st.txt > $OUT/bench_shapes_$st.log 2>&1
  head -12 $OUT/r05_gemm_shapes_in_step_stagger$st.txt
done
timeout 1800 python3 -m pytest tests/test_hip_bf16_stages.py tests/test_hip_configs.py tests/test_hip_model.py tests/test_hip_swin.py -m gpu -x -q -s -k "train_mode_weight or config4_full_size or fused_adam_steps or gradient_link or middle_pair" > $OUT/pytest_new.log 2>&1
grep -v "Warning\|warn" $OUT/pytest_new.log | grep "train-mode TswinPlus\|configs\[4\] fp8-step\|losses fused\|passed\|failed\|Error" | head
