#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05q
python3 bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | tee gpurun_out/r05q/r05_soak_4000_steps_final.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('soak 4000', d['value'], d['ms_per_step'], d['config'].get('loss'))"
python3 bench.py --workload contrast --steps 400 --warmup 3 2>/dev/null | grep '^{"metric"' | tee gpurun_out/r05q/r05_soak_contrast_400_steps_final.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('soak contrast 400', d['value'], d['ms_per_step'], d['config'].get('loss'))"
python3 bench.py --workload contrast --bank batch --steps 100 --warmup 3 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('contrast bank 100', d['value'], d['ms_per_step'], d['config'].get('loss'))"
