#!/bin/bash
# kernel trace of the bench step -> ordered launch list of one step (gpurun_out/$1_step_sequence.txt); $2 = 0 eager (default) | 1 graph replay
TAG=${1:-r06}
GRAPH=${2:-0}
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_seq
cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace -d /tmp/prof_seq -o kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-profile --no-calibration --graph $GRAPH > gpurun_out/${TAG}_seq_run.log 2>&1
KT=$(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1)
python3 tools/step_sequence.py "$KT" > gpurun_out/${TAG}_step_sequence.txt 2>&1
grep '^{"metric"' gpurun_out/${TAG}_seq_run.log | cut -c 1-200
head -3 gpurun_out/${TAG}_step_sequence.txt
grep "^# step" gpurun_out/${TAG}_step_sequence.txt
