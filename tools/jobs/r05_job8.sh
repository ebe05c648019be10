#!/bin/bash
# round 5, GPU job 8: product stagger - parity, per-shape sweep behind a spacer, in-step A/B; bank kernel LSE (unit rows); fixed tests
set -u
OUT=gpurun_out/r05h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python3 -m pytest tests/test_hip_gemm.py tests/test_hip_production_dispatch.py tests/test_hip_contrast_bank.py tests/test_hip_bf16_stages.py tests/test_hip_configs.py tests/test_hip_model.py tests/test_hip_swin.py -m gpu -x -q -s -k "not bench_two_ranks" > $OUT/pytest.log 2>&1
grep -v "Warning\|warn" $OUT/pytest.log | grep "train-mode TswinPlus\|configs\[4\] fp8-step\|losses fused\|passed\|failed\|Error" | head
timeout 900 python3 tools/stagger_sweep.py > $OUT/r05_stagger_sweep.txt 2>&1
grep -v amdgpu.ids $OUT/r05_stagger_sweep.txt
for rep in 1 2 3; do
  for st in 0 1; do
    STSWIN_NT_STAGGER=$st timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_stagger${st}_$rep.log 2>&1
    echo "STSWIN_NT_STAGGER=$st rep $rep: $(grep '^{"metric"' $OUT/bench_stagger${st}_$rep.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "frames/s", round(d["ms_per_step"],3), "ms", round(d["roofline"]["frac"],4))')" | tee -a $OUT/r05_stagger_in_step_ab.txt
  done
done
timeout 600 python3 tools/bench_contrast.py > $OUT/r05_contrast_kernels.txt 2>&1
grep -v amdgpu.ids $OUT/r05_contrast_kernels.txt
