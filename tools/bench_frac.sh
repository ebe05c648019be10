#!/bin/bash
# bench.py at several step counts: frames/s, gemm_nt roofline fraction and the other timed kernel families (run through gpurun).
for st in "$@"; do
  python3 bench.py --steps $st --warmup 4 --no-cpu-baseline --no-secondary 2>/dev/null | grep '^{"metric"' | python3 -c '
import json, sys
d = json.loads(sys.stdin.read()); r = d["roofline"]
print(d["steps"], "steps", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 3), "ms  gemm_nt frac", round(r["frac"], 4),
      "launches/step", r["launches_per_step"], "timed", r["launches_timed"])
print("   ", {k: (round(v["ms_per_step"], 3), round(v["tflops"], 1)) for k, v in r["other_kernels"].items()})'
done
