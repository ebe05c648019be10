#!/bin/bash
# round 5, GPU job 7: new parity tests (train-mode whole-model gradients, fp8-vs-bf16 step gradients, tightened bounds), then the
# stagger experiment behind a spacer kernel (tuning build)
set -u
OUT=gpurun_out/r05g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests/test_hip_bf16_stages.py tests/test_hip_configs.py tests/test_hip_model.py tests/test_hip_swin.py -m gpu -x -q -s -k "train_mode_weight or config4_full_size or fused_adam_steps or gradient_link or middle_pair" > $OUT/pytest_new.log 2>&1
grep -v "Warning\|warn" $OUT/pytest_new.log | tail -25
STSWIN_TUNING=1 timeout 900 python3 __graft_entry__.py --force > $OUT/build_tuning.log 2>&1 || tail -20 $OUT/build_tuning.log
timeout 900 python3 tools/stagger_ab2.py > $OUT/r05_stagger_behind_spacer.txt 2>&1
grep -v amdgpu.ids $OUT/r05_stagger_behind_spacer.txt
