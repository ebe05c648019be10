#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV over the steady-state window only.

usage: prof_summary.py <kernel_trace.csv> [--last-ms MS | --last-frac F] [--top N]
Kernels are grouped by (shortened) name; the window is the last MS milliseconds (or fraction F) of the trace, so
first-call library autotuning (MIOpen find) and warm-up steps do not pollute the table."""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_Z\d+(\w+?)I", name)
    if name.startswith("_Z"):
        import subprocess
        try:
            name = subprocess.run(["c++filt", name.split("(")[0]], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
    name = re.sub(r"<.*", "", name) if len(name) > 90 else name
    return name[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--last-ms", type=float, default=None)
    ap.add_argument("--last-frac", type=float, default=None)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--gaps", type=int, default=0, help="also list the N kernels that the longest idle time sits in front of")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t_end = max(r[1] for r in rows)
    t_begin = min(r[0] for r in rows)
    if a.last_ms is not None:
        t0 = t_end - int(a.last_ms * 1e6)
    elif a.last_frac is not None:
        t0 = t_end - int((t_end - t_begin) * a.last_frac)
    else:
        t0 = t_begin
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, n in rows:
        if s < t0:
            continue
        k = short(n)
        agg[k][0] += 1
        agg[k][1] += e - s
        busy += e - s
    span = t_end - t0
    print(f"# window {span / 1e6:.2f} ms, kernel-busy {busy / 1e6:.2f} ms ({100.0 * busy / span:.1f}%), {sum(v[0] for v in agg.values())} launches")
    print(f"{'kernel':92s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'%busy':>6s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: a.top]:
        print(f"{k:92s} {c:6d} {t / 1e6:9.3f} {t / c / 1e3:9.1f} {100.0 * t / busy:6.2f}")
    if a.gaps:
        # idle time of the device in front of each launch (start minus the latest end seen so far), charged to the kernel that
        # follows the gap AND to the one that precedes it: where the host does not keep up
        gap_after, gap_before = defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0])
        last_end, last_name, idle = None, None, 0
        for s, e, n in rows:
            if s < t0:
                last_end, last_name = max(last_end or e, e), short(n)
                continue
            if last_end is not None and s > last_end:
                g = s - last_end
                idle += g
                gap_before[short(n)][0] += 1
                gap_before[short(n)][1] += g
                gap_after[last_name][0] += 1
                gap_after[last_name][1] += g
            if last_end is None or e > last_end:
                last_end, last_name = e, short(n)
        print(f"\n# device idle inside the window: {idle / 1e6:.3f} ms ({100.0 * idle / span:.1f}%)")
        print(f"{'idle in FRONT of kernel':92s} {'gaps':>6s} {'idle_ms':>9s} {'avg_us':>9s}")
        for k, (c, t) in sorted(gap_before.items(), key=lambda kv: -kv[1][1])[: a.gaps]:
            print(f"{k:92s} {c:6d} {t / 1e6:9.3f} {t / c / 1e3:9.1f}")
        print(f"{'idle BEHIND kernel':92s} {'gaps':>6s} {'idle_ms':>9s} {'avg_us':>9s}")
        for k, (c, t) in sorted(gap_after.items(), key=lambda kv: -kv[1][1])[: a.gaps]:
            print(f"{k:92s} {c:6d} {t / 1e6:9.3f} {t / c / 1e3:9.1f}")


if __name__ == "__main__":
    sys.exit(main())
