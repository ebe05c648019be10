#!/bin/bash
# round 5, GPU job 2: the self-pipelined duo GEMM (parity + speed per epilogue) and the scratch-free stage-2 attention backward
set -u
OUT=gpurun_out/r05b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_hip_gemm.py tests/test_hip_attention.py tests/test_hip_fp8.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 900 python3 tools/epi_sweep.py > $OUT/r05_epi_sweep_duo.txt 2>&1
cat $OUT/r05_epi_sweep_duo.txt
timeout 600 python3 tools/bench_attn.py > $OUT/r05_attention_kernels.txt 2>&1
cat $OUT/r05_attention_kernels.txt
