#!/bin/bash
# round 5, GPU job 28: whole -m gpu suite on the final tree + a 1500-step soak of the bench (loss finite, step time stable) + smoke
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2700 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all_final.log 2>&1
tail -4 $OUT/pytest_all_final.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | tee $OUT/r05_soak_1500_steps.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('soak', d['value'], d['ms_per_step'], d['config'].get('loss'))"
python3 bench.py --workload contrast --steps 300 --warmup 3 2>/dev/null | grep '^{"metric"' | tee $OUT/r05_soak_contrast_300_steps.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('soak contrast', d['value'], d['ms_per_step'], d['config'].get('loss'))"
