#!/bin/bash
# Diagnosis builds of the 4-wave register-pipelined ring kernel (run HERE, before gpurun): the library once per removed instruction
# stream (wrong results - the MFMA stream is unchanged), for tools/probes/w4_loop_cost.py.
set -e
cd "$(dirname "$0")/../.."
mkdir -p stswincl_amd/lib/ab
for v in ${W4X_VARIANTS:-BASE NO_DMA NO_READ NO_BARRIER NO_WAIT NO_DMA_NO_READ}; do
  defs=""
  case $v in
    NO_DMA) defs="-DSTSWIN_W4X_NO_DMA";; NO_READ) defs="-DSTSWIN_W4X_NO_READ";; NO_BARRIER) defs="-DSTSWIN_W4X_NO_BARRIER";;
    NO_WAIT) defs="-DSTSWIN_W4X_NO_WAIT";; SCHED1) defs="-DSTSWIN_W4X_SCHED=1";; SCHED2) defs="-DSTSWIN_W4X_SCHED=2";; SCHED3) defs="-DSTSWIN_W4X_SCHED=3";; NO_DMA_NO_READ) defs="-DSTSWIN_W4X_NO_DMA -DSTSWIN_W4X_NO_READ";;
  esac
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=100000 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -DSTSWIN_TUNING $defs $W4X_EXTRA -I include \
      -c stswincl_amd/csrc/gemm.hip -o stswincl_amd/lib/ab/gemm_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o stswincl_amd/lib/ab/libstswin_w4_$v.so stswincl_amd/lib/ab/gemm_$v.o \
      stswincl_amd/lib/attention.o stswincl_amd/lib/contrast.o stswincl_amd/lib/conv_halo.o stswincl_amd/lib/headops.o stswincl_amd/lib/optim.o \
      $(ls stswincl_amd/lib/*.o | grep -v "gemm.o\|attention.o\|contrast.o\|conv_halo.o\|headops.o\|optim.o") && rm stswincl_amd/lib/ab/gemm_$v.o ) &
done
wait
ls -la stswincl_amd/lib/ab/
