#!/bin/bash
# round 5, GPU job 18: LayerNorm backward (gamma from LDS, template switches, 8-wave workgroups) - tests, kernel table, step A/B vs round-4 tree
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_hip_rowops.py tests/test_hip_swin.py tests/test_hip_bf16_stages.py -m gpu -x -q 2>&1 | tail -2
{
echo "## round-4 library"; STSWIN_HIP_LIB=$PWD/.ab_old/r04/stswincl_amd/lib/libstswin_hip.so python3 tools/bench_ln.py
echo "## this tree"; python3 tools/bench_ln.py
echo "## this tree, 4-wave workgroups everywhere"; STSWIN_LN_BWD_WAVES=4 python3 tools/bench_ln.py
} 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_layernorm_kernels.txt
git_head=none
for i in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export STSWIN_HIP_LIB=$PWD/.ab_old/r05pre/libstswin_hip.so; else unset STSWIN_HIP_LIB; fi
    python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms')"
  done
done 2>&1 | tee $OUT/r05_layernorm_in_step_ab.txt
