#!/bin/bash
# round 5, GPU job 16: whole -m gpu suite on the documented tree + the driver's default bench line + smoke
set -u
OUT=gpurun_out/r05p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2700 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.log 2>&1
tail -6 $OUT/pytest_all.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py 2>/dev/null | grep '^{"metric"' > $OUT/bench_default.json
python3 -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d.get('secondary',{}).get('value'), d.get('secondary',{}).get('bank',{}).get('value'))"
