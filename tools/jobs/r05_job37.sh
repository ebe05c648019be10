#!/bin/bash
# round 5, GPU job 37: whole -m gpu suite on the tree with the grouped weight gradients + 2-rank functional bench on one GPU + soak
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2700 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all_final.log 2>&1
grep -a "passed\|failed" $OUT/pytest_all_final.log | tail -2
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
STSWIN_BENCH_SHARE_GPU=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --batch 2 --steps 4 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2 ranks on one GPU', d['value'], d['ms_per_step'], d['config'].get('launch'), d['config'].get('loss'))"
python3 bench.py --steps 600 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | tee $OUT/r05_soak_600_steps_grouped.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('soak', d['value'], d['ms_per_step'], d['config'].get('loss'))"
python3 bench.py --workload contrast --steps 100 --warmup 3 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('contrast', d['value'], d['ms_per_step'], d['config'].get('loss'))"
