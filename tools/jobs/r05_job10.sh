#!/bin/bash
# round 5, GPU job 10: in-step A/B of the fused QKV + attention forward in front of a backward, and of fp8-stored q | k | v; new tests
set -u
OUT=gpurun_out/r05j
mkdir -p $OUT
export TMPDIR=/tmp
line() { grep '^{"metric"' $1 | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "frames/s", round(d["ms_per_step"],3), "ms", round(d["roofline"]["frac"],4))'; }
for rep in 1 2 3; do
  for v in "" "STSWIN_FUSED_QKV=1" "STSWIN_FP8_ATTN=1"; do
    env $v timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/b.log 2>&1
    echo "[${v:-default}] rep $rep: $(line $OUT/b.log)" | tee -a $OUT/r05_fusedqkv_fp8_in_step_ab.txt
  done
done
timeout 1800 python3 -m pytest tests/test_hip_bf16_stages.py tests/test_hip_configs.py tests/test_hip_model.py tests/test_hip_swin.py tests/test_hip_contrast_bank.py -m gpu -x -q -s -k "train_mode_weight or config4_full_size or fused_adam_steps or gradient_link or middle_pair or production_size_bank" > $OUT/pytest_new.log 2>&1
grep -v "Warning\|warn" $OUT/pytest_new.log | grep "train-mode TswinPlus\|configs\[4\] fp8-step\|losses fused\|passed\|failed\|Error" | head
