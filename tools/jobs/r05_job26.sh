#!/bin/bash
# round 5, GPU job 26: steady-state kernel table of the step after the BatchNorm / LayerNorm changes + contrastive step
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_under_rocprof.log 2>&1
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python3 tools/prof_summary.py "$KT" --last-ms 150 --top 45 --gaps 5 > $OUT/steady_mid_round.txt 2>&1
head -50 $OUT/steady_mid_round.txt
python3 bench.py --workload contrast --steps 8 --warmup 3 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('contrast', d['value'], d['ms_per_step'])"
python3 bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('seg', d['value'], d['ms_per_step'], d['roofline']['frac'])"
