#!/bin/bash
# round 5, GPU job 6: whole -m gpu suite on the tree, then same-box A/B of the round-4 tree (.ab_old/r04) against this one
set -u
OUT=gpurun_out/r05f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1
tail -6 $OUT/pytest_all.log
bash tools/ab_trees.sh .ab_old/r04 3 2>&1 | tee $OUT/r05_ab_vs_round4.txt
for sw in STSWIN_NO_BIAS_CACHE=1 STSWIN_TORCH_PADVEC=1; do
  env $sw python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$sw', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')" | tee -a $OUT/r05_ab_vs_round4.txt
done
