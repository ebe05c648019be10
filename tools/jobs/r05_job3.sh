#!/bin/bash
# round 5, GPU job 3: duo parity (all K tails), attention / fp8 suites on the scratch-free kernels, in-step A/B of the duo sites
set -u
OUT=gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_hip_gemm.py tests/test_hip_attention.py tests/test_hip_fp8.py tests/test_hip_swin.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
for rep in 1 2; do
  for duo in 0 fc1 fc1,fc2d fc1,fc1ng; do
    STSWIN_DUO=$duo timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_duo_${duo}_$rep.log 2>&1
    echo "STSWIN_DUO=$duo rep $rep: $(grep '^{"metric"' $OUT/bench_duo_${duo}_$rep.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "frames/s", round(d["ms_per_step"],3), "ms", d["roofline"]["frac"])')" | tee -a $OUT/r05_duo_in_step_ab.txt
  done
done
timeout 600 python3 tools/bench_attn_qkv.py > $OUT/r05_attention_qkv_fused_bench.txt 2>&1
cat $OUT/r05_attention_qkv_fused_bench.txt
