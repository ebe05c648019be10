#!/bin/bash
# round 5, GPU job 17: LayerNorm backward row-loop forms - parity tests, then the round-4 library vs variants A..E (STSWIN_LN_BWD_VAR)
set -u
OUT=gpurun_out/r05q
mkdir -p $OUT
export TMPDIR=/tmp
for v in A C E; do STSWIN_LN_BWD_VAR=$v timeout 900 python3 -m pytest tests/test_hip_rowops.py -m gpu -x -q -k "layernorm or ln" 2>&1 | tail -1; done
{
echo "## round-4 library"; STSWIN_HIP_LIB=$PWD/.ab_old/r04/stswincl_amd/lib/libstswin_hip.so python3 tools/bench_ln.py
for v in A B C D E; do
  echo "## this tree, variant $v (A keep/global gamma, B keep/LDS gamma, C B + fma, D re-derive/LDS/fma, E D with 2 rows in flight)"
  STSWIN_LN_BWD_VAR=$v python3 tools/bench_ln.py
done
echo "## this tree, variant C, 4 waves"; STSWIN_LN_BWD_VAR=C STSWIN_LN_BWD_WAVES=4 python3 tools/bench_ln.py
} 2>&1 | grep -v amdgpu.ids | tee $OUT/r05_layernorm_kernels.txt
