#!/bin/bash
# round 5, GPU job 1: gemm tests of the ticketed fused combine, a bench line of the tree, then a STSWIN_TUNING build for the in-kernel
# timeline of the epilogue-heavy MLP tiles (verdict item 1: "first commit an in-kernel timeline of one tile") and the variant sweep.
set -u
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_hip_gemm.py tests/test_hip_production_dispatch.py tests/test_hip_abi.py -m gpu -x -q > $OUT/pytest_gemm.log 2>&1
tail -5 $OUT/pytest_gemm.log
timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_a.log 2>&1
grep '^{"metric"' $OUT/bench_a.log | tail -1 | cut -c1-400
STSWIN_TN_FUSED=0 timeout 600 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/bench_unfused.log 2>&1
grep '^{"metric"' $OUT/bench_unfused.log | tail -1 | cut -c1-200
# tuning build (same path: the box is scratch)
STSWIN_TUNING=1 timeout 900 python3 __graft_entry__.py --force > $OUT/build_tuning.log 2>&1 || tail -20 $OUT/build_tuning.log
for epi in plain gelu_dgelu mul_r; do
  for shape in "65536 2048 512" "16384 4096 1024"; do
    STSWIN_TL_EPI=$epi timeout 300 python3 tools/gemm_timeline.py $shape >> $OUT/r05_gemm_tile_timeline_baseline.txt 2>&1
  done
done
cat $OUT/r05_gemm_tile_timeline_baseline.txt
timeout 900 python3 tools/epi_sweep.py > $OUT/r05_epi_sweep_variants_baseline.txt 2>&1
cat $OUT/r05_epi_sweep_variants_baseline.txt
