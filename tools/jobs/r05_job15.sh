#!/bin/bash
set -u
OUT=gpurun_out/r05o
mkdir -p $OUT
export TMPDIR=/tmp
echo "== BN_BWD_U = 4 (product)" > $OUT/r05_bn_bwd_rows_in_flight.txt
timeout 300 python3 tools/bench_bn.py 2>&1 | grep -v amdgpu.ids >> $OUT/r05_bn_bwd_rows_in_flight.txt
echo "== BN_BWD_U = 8 (eight rows per thread in flight)" >> $OUT/r05_bn_bwd_rows_in_flight.txt
STSWIN_HIP_LIB=$PWD/stswincl_amd/lib/u8/libstswin_hip.so timeout 300 python3 tools/bench_bn.py 2>&1 | grep -v amdgpu.ids >> $OUT/r05_bn_bwd_rows_in_flight.txt
cat $OUT/r05_bn_bwd_rows_in_flight.txt
for rep in 1 2; do
  for lib in "" "STSWIN_HIP_LIB=$PWD/stswincl_amd/lib/u8/libstswin_hip.so"; do
    env $lib timeout 600 python3 bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[${lib:+U=8}] ', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')" | tee -a $OUT/r05_bn_bwd_rows_in_flight.txt
  done
done
