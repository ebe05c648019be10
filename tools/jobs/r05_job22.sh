#!/bin/bash
# round 5, GPU job 22: BatchNorm kernels with LDS-distributed channel constants (apply, bwd reduce, bwd dx at 2048 workgroups) - tests, table, step A/B
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_hip_head.py tests/test_hip_syncbn.py tests/test_hip_primitives.py tests/test_hip_model.py tests/test_hip_configs.py -m gpu -x -q 2>&1 | tail -2
python3 tools/bench_bn.py 2>&1 | grep -v "amdgpu.ids" | tee $OUT/r05_batchnorm_kernels_lds_constants.txt
for i in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export STSWIN_HIP_LIB=$PWD/.ab_old/r05pre/libstswin_hip.so; else unset STSWIN_HIP_LIB; fi
    python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
  done
done 2>&1 | tee $OUT/r05_bn_lds_constants_in_step_ab.txt
