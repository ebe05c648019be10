#!/bin/bash
# Diagnosis: what do the LDS fragment reads of the 256x256 ring gemm_nt cost on RANDOM operands (where the loop is clock-limited)?
# Builds a second library whose main loop issues HALF the ds_read_b128 (the other fragments are register copies: wrong results, the
# same MFMA stream on live data) and times the yardstick shapes with both.  Run on the GPU box from the repo root.
set -e
L=stswincl_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -DSTSWIN_DEBUG_HALF_READS -I include -c stswincl_amd/csrc/gemm.hip -o /tmp/gemm_half.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libstswin_half.so /tmp/gemm_half.o $L/rowops.o $L/attention.o $L/headops.o $L/contrast.o $L/optim.o $L/conv_halo.o $L/selftest.o
echo "== normal build"; python3 tools/blas_compare.py 2>&1 | grep -v amdgpu
echo "== half the fragment reads (results are wrong by construction)"; STSWIN_HIP_LIB=/tmp/libstswin_half.so python3 tools/blas_compare.py 2>&1 | grep -v amdgpu
echo "== normal build, zero operands"; ZERO=1 python3 tools/blas_compare.py 2>&1 | grep -v amdgpu
