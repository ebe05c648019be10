#!/bin/bash
# Same-box A/B of two source trees (run through gpurun): tools/ab_trees.sh OLD_DIR [rounds] [bench args...] -> frames/s of each run,
# alternating old / new (OLD_DIR = a built checkout of an earlier commit inside the repo, e.g. a git worktree under .ab_old/).
OLD=$1; R=${2:-2}; shift 2
for i in $(seq 1 $R); do
  for side in old new; do
    if [ $side = old ]; then D=$OLD; else D=.; fi
    (cd $D && python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('$side', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms  gemm_nt frac', round(r.get('frac',0),4))")
  done
done
