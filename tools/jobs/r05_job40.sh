#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05q
python3 tools/host_time.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05q/r05_host_enqueue_time.txt
