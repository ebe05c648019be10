#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean counter value per dispatch (steady-state tail)."""
import csv, sys, re
from collections import defaultdict
path, counter = sys.argv[1], sys.argv[2]
skip_frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = list(csv.DictReader(open(path)))
rows = [r for r in rows if r.get("Counter_Name") == counter]
rows = rows[int(len(rows) * skip_frac):]
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    k = r["Kernel_Name"]
    k = re.sub(r"^void ", "", k)[:60]
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
print(f"# {counter}: mean per dispatch over the last {100 * (1 - skip_frac):.0f}% of dispatches")
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k:62s} n={n:5d} mean={v / n:14.1f} total={v:16.1f}")
