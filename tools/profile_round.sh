#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root): rocprofv3 kernel trace + stats of the default bench
# command, the steady-state per-kernel table, the per-shape GEMM table (live HIP events), PMC passes (FETCH_SIZE / WRITE_SIZE,
# separate runs, no tracing flags beside --kernel-trace) and the contrastive workload.  Summaries land in gpurun_out/$1/.
set -u
TAG=${1:-r03}
COMMIT=${2:-unknown}        # the commit this tree was snapshotted from (.git does not travel to the GPU box): `git rev-parse --short HEAD`
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
run_prof() {  # name, rocprof flags..., -- cmd
  local name=$1; shift
  local dir=/tmp/prof_$name
  rm -rf $dir
  rocprofv3 "$@" > $OUT/${name}_run.log 2>&1
  echo "$dir"
}
# 1. kernel trace + stats
rm -rf /tmp/prof_kt
# (round 6: the timed steps are hipGraph replays; no bracketed pass and no calibration probes inside the trace, so that the last 150 ms are replayed steps only)
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_kt -o kt -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --no-profile --no-calibration > $OUT/bench_under_rocprof.log 2>&1
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -60 "$ST" > $OUT/${TAG}_rocprofv3_kernel_stats.csv
[ -n "$KT" ] && python3 tools/prof_summary.py "$KT" --last-ms 150 --top 60 --gaps 25 > $OUT/${TAG}_bench_steady_state_kernels.txt 2>&1
[ -n "$KT" ] && python3 tools/step_sequence.py "$KT" --no-list > $OUT/${TAG}_step_sequence_graph.txt 2>&1
grep '^{"metric"' $OUT/bench_under_rocprof.log | tail -1 > $OUT/${TAG}_bench_line_under_rocprof.json
# 2. clean bench line + per-shape table
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_clean.log 2>&1
grep '^{"metric"' $OUT/bench_clean.log | tail -1 > $OUT/${TAG}_bench_line.json
STSWIN_SHAPE_PROFILE=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --profile-stride 1 --dump-prof $OUT/${TAG}_gemm_shapes_in_step.txt > $OUT/bench_shapes.log 2>&1
# 3. PMC passes (each its own run)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc_$C
  rocprofv3 --output-format csv --pmc $C --kernel-trace -d /tmp/prof_pmc_$C -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-calibration --no-secondary --graph 0 > $OUT/pmc_$C.log 2>&1
  CC=$(find /tmp/prof_pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$CC" ] && python3 tools/pmc_summary.py "$CC" $C 0.34 > $OUT/${TAG}_pmc_$(echo $C | tr A-Z a-z).txt 2>&1
  [ -n "$CC" ] && cp "$CC" /tmp/pmc_$C.csv
done
[ -f /tmp/pmc_FETCH_SIZE.csv ] && [ -f /tmp/pmc_WRITE_SIZE.csv ] && python3 tools/pmc_dominant.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv $OUT/${TAG}_pmc_dominant_kernel.json $COMMIT > $OUT/pmc_dominant.log 2>&1
# 4. contrastive workload
python3 bench.py --workload contrast --steps 6 --warmup 2 > $OUT/bench_contrast.log 2>&1
grep '^{"metric"' $OUT/bench_contrast.log | tail -1 > $OUT/${TAG}_bench_contrast_line.json
rm -rf /tmp/prof_ktc
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_ktc -o kt -- python3 bench.py --workload contrast --steps 4 --warmup 2 --no-profile --graph 0 > $OUT/bench_contrast_under_rocprof.log 2>&1
KTC=$(find /tmp/prof_ktc -name "*kernel_trace.csv" | head -1)
[ -n "$KTC" ] && python3 tools/prof_summary.py "$KTC" --last-ms 120 --top 50 --gaps 15 > $OUT/${TAG}_contrast_steady_state_kernels.txt 2>&1
python3 bench.py --workload contrast --bank batch --steps 6 --warmup 2 > $OUT/bench_contrast_bank.log 2>&1
grep '^{"metric"' $OUT/bench_contrast_bank.log | tail -1 > $OUT/${TAG}_bench_contrast_bank_line.json
python3 bench.py --batch 8 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/bench_b8.log 2>&1
grep '^{"metric"' $OUT/bench_b8.log | tail -1 > $OUT/${TAG}_bench_batch8_line.json
# 5. kernel micro-benchmarks
python3 tools/bench_attn.py > $OUT/${TAG}_attention_kernels.txt 2>&1
python3 tools/bench_contrast.py > $OUT/${TAG}_contrast_kernels.txt 2>&1
python3 tools/attn_timeline8.py > $OUT/${TAG}_attn_bwd8_timeline.txt 2>&1
python3 tools/bench_attn_qkv.py > $OUT/${TAG}_attention_qkv_fused_bench.txt 2>&1
python3 tools/attn_qkv_timeline.py > $OUT/${TAG}_attention_qkv_fused_timeline.txt 2>&1
python3 tools/bench_aspp_taps.py > $OUT/${TAG}_aspp_tap_skipping.txt 2>&1
python3 tools/bench_bn.py > $OUT/${TAG}_batchnorm_kernels.txt 2>&1
python3 tools/bench_conv_halo.py > $OUT/${TAG}_conv_halo_bench.txt 2>&1
python3 tools/conv_halo_timeline.py > $OUT/${TAG}_conv_halo_timeline_fwd.txt 2>&1
python3 tools/conv_halo_timeline.py --dgrad > $OUT/${TAG}_conv_halo_timeline_dgrad.txt 2>&1
python3 tools/conv_halo_timeline.py --wgrad > $OUT/${TAG}_conv_halo_timeline_wgrad.txt 2>&1
python3 tools/stem_s2d_experiment.py > $OUT/${TAG}_stem_kernels.txt 2>&1
python3 tools/bench_stem_tail.py > $OUT/${TAG}_stem_tail.txt 2>&1
python3 tools/blas_compare.py > $OUT/${TAG}_vendor_blas_yardstick.txt 2>&1
python3 tools/torch_ops.py > $OUT/${TAG}_torch_ops.txt 2>&1
python3 -m pytest tests/test_hip_bf16_stages.py -q -s 2>&1 | grep -E "^\.?(swin|patch|conv|ASPP|eval)" > $OUT/${TAG}_bf16_stage_parity_table.txt
ls -la $OUT
