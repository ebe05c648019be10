#!/bin/bash
# round 5, GPU job 41: overlap proxy again, with the grouped weight-gradient launch on and off
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r05q
{ echo "## grouped weight gradients ON (default)"; python3 tools/overlap_proxy.py; echo "## grouped weight gradients OFF (STSWIN_TN_GROUP=0)"; STSWIN_TN_GROUP=0 python3 tools/overlap_proxy.py; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05q/r05_overlap_proxy_grouped.txt
