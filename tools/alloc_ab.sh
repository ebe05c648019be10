#!/bin/bash
# does the caching allocator's segment layout move the step? (same box, alternating)
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-profile --no-calibration 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2; do
  run default
  PYTORCH_HIP_ALLOC_CONF=expandable_segments:True PYTORCH_CUDA_ALLOC_CONF=expandable_segments:True run expandable
  PYTORCH_HIP_ALLOC_CONF=max_split_size_mb:8192 PYTORCH_CUDA_ALLOC_CONF=max_split_size_mb:8192 run bigsplit
done
