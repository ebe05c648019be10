#!/bin/bash
set -u
export TMPDIR=/tmp
for a in "--graph 1 --no-profile" "--graph 0 --no-profile" "--batch 8 --graph 1 --no-profile"; do
python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary $a 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms', d['config'].get('launch'), d['config'].get('loss'))"
done
STSWIN_FP8_ATTN=1 python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-secondary 2>&1 | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp8 attention:', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms', d['config'].get('loss'))"
