#!/bin/bash
set -u
OUT=gpurun_out/r05n
mkdir -p $OUT
export TMPDIR=/tmp
for r in 512 1024 2048 4096; do for d in 512 1024 2048 4096; do
  [ $r != 512 ] && [ $d != 512 ] && [ $r != $d ] && continue
  echo "== reduce target $r workgroups, dx target $d" >> $OUT/r05_bn_bwd_workgroup_sweep.txt
  STSWIN_BN_RED_WGS=$r STSWIN_BN_DX_WGS=$d timeout 300 python3 tools/bench_bn.py 2>&1 | grep -v amdgpu.ids >> $OUT/r05_bn_bwd_workgroup_sweep.txt
done; done
cat $OUT/r05_bn_bwd_workgroup_sweep.txt
