#!/bin/bash
# A/B kernel tables of the default bench command under an environment switch (run through gpurun):
#   tools/ab_prof.sh TAG VAR=VALUE [bench args...]  ->  gpurun_out/TAG/{a,b}_kernels.txt  (a = default, b = with the switch)
set -u
TAG=$1; SW=$2; shift 2
for arg in "$@"; do case "$arg" in --gpus*) echo "ab_prof.sh: profiled runs are single-GPU (bench.py would become a launcher under rocprofv3)"; exit 2;; esac; done
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for side in a b; do
  rm -rf /tmp/prof_$side
  if [ $side = b ]; then export "$SW"; fi
  rocprofv3 --output-format csv --kernel-trace -d /tmp/prof_$side -o kt -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-profile "$@" > $OUT/${side}_run.log 2>&1
  KT=$(find /tmp/prof_$side -name "*kernel_trace.csv" | head -1)
  python3 tools/prof_summary.py "$KT" --last-ms 150 --top 70 > $OUT/${side}_kernels.txt 2>&1
  tail -1 $OUT/${side}_run.log | cut -c1-260
done
