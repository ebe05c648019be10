#!/bin/bash
# round 5, GPU job 5: contrastive token path (tests, bench A/B, remaining torch launches), stagger re-measurement (tuning build)
set -u
OUT=gpurun_out/r05e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_hip_contrast.py tests/test_hip_contrast_bank.py tests/test_hip_syncbn.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
for rep in 1 2; do
  for glue in 1 0; do
    STSWIN_CONTRAST_TORCH_GLUE=$glue timeout 600 python3 bench.py --workload contrast --steps 20 --warmup 3 > $OUT/bench_contrast_glue${glue}_$rep.log 2>&1
    echo "torch glue=$glue rep $rep: $(grep '^{"metric"' $OUT/bench_contrast_glue${glue}_$rep.log | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["unit"], round(d["ms_per_step"],3), "ms")')" | tee -a $OUT/r05_contrast_glue_ab.txt
  done
done
timeout 600 python3 tools/torch_ops.py contrast > $OUT/r05_torch_ops_contrast.txt 2>&1
grep -v "Warning\|warn\|amdgpu.ids" $OUT/r05_torch_ops_contrast.txt | head -60
