#!/usr/bin/env python3
"""One training step of a rocprofv3 --kernel-trace CSV as an ORDERED launch list (round 6: the pass-and-launch diet).

usage: step_sequence.py <kernel_trace.csv> [--marker stem_s2d] [--step -2]
A step = the launches from one occurrence of the marker kernel (the first kernel of the forward pass) up to the next.  Prints the
launch count of every complete step in the trace (the counting method of DESIGN's "launches per step": device dispatches between two
consecutive first kernels, Adam and re-packing included), then the chosen step as `index  start_us  dur_us  gap_us  kernel`, and a
per-kernel table of it (calls, total us; GEMM families marked) with the non-GEMM total."""
import argparse
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import short  # noqa: E402

GEMM = ("gemm_nt", "gemm_tn", "conv3x3_c64", "stem_conv", "stem_wgrad", "attn_", "tn_reduce", "nt_splitk")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--marker", default="stem_s2d")
    ap.add_argument("--step", type=int, default=-2, help="which complete step to list (python index; -1 = the last complete one)")
    ap.add_argument("--no-list", action="store_true")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    # the marker may run several times per step (once per forward): a step starts at a marker that follows an optimizer kernel
    starts = [i for i in marks if i == marks[0] or any("multi_tensor" in rows[j][2] for j in range(max(0, i - 40), i))]
    steps = [(starts[k], starts[k + 1]) for k in range(len(starts) - 1)]
    if not steps:
        print("no complete step found")
        return
    print("# launches per complete step: " + ", ".join(str(e - s) for s, e in steps))
    print("# step time (first kernel to next step's first kernel), ms: " + ", ".join(f"{(rows[e][0] - rows[s][0]) / 1e6:.2f}" for s, e in steps))
    s, e = steps[a.step]
    seq = rows[s:e]
    t0 = seq[0][0]
    agg = defaultdict(lambda: [0, 0])
    prev_end = t0
    gaps = 0
    for i, (st, en, n) in enumerate(seq):
        k = short(n)
        agg[k][0] += 1
        agg[k][1] += en - st
        gap = max(0, st - prev_end)
        gaps += gap
        if not a.no_list:
            print(f"{i:4d} {(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.1f} {gap / 1e3:7.1f}  {k}")
        prev_end = max(prev_end, en)
    busy = sum(v[1] for v in agg.values())
    gemm = sum(v[1] for k, v in agg.items() if any(g in k for g in GEMM))
    print(f"# step {a.step}: {len(seq)} launches, kernel time {busy / 1e6:.3f} ms (GEMM / conv / attention families {gemm / 1e6:.3f} ms, "
          f"everything else {(busy - gemm) / 1e6:.3f} ms), idle between kernels {gaps / 1e6:.3f} ms")
    print(f"{'kernel':92s} {'calls':>6s} {'total_us':>9s} {'avg_us':>8s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:92s} {c:6d} {t / 1e3:9.1f} {t / 1e3 / c:8.1f}{'   *' if any(g in k for g in GEMM) else ''}")


if __name__ == "__main__":
    main()
