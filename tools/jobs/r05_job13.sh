#!/bin/bash
set -u
OUT=gpurun_out/r05m
mkdir -p $OUT
export TMPDIR=/tmp
STSWIN_TUNING=1 timeout 900 python3 __graft_entry__.py --force > $OUT/build_tuning.log 2>&1 || tail -20 $OUT/build_tuning.log
timeout 900 python3 tools/persist_ab.py > $OUT/r05_persistent_tile_loop.txt 2>&1
grep -v amdgpu.ids $OUT/r05_persistent_tile_loop.txt
for rep in 1 2; do for v in 0 1; do
  STSWIN_NT_PERSIST=$v timeout 600 python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('STSWIN_NT_PERSIST=$v (tuning build)', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')" | tee -a $OUT/r05_persistent_tile_loop.txt
done; done
