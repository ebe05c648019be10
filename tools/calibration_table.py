#!/usr/bin/env python3
"""Collect the `calibration` blocks of bench lines (gpurun_out/*.json, profiles/*bench_line*.json) into one table: one row per run with
value, probes and value_normalised - the source of bench.py's CAL_REF (medians) and of profiles/r06_calibration_boxes.txt."""
import glob
import json
import statistics
import sys

rows = []
for path in sorted(set(glob.glob("gpurun_out/**/*.json", recursive=True) + glob.glob("gpurun_out/**/*.log", recursive=True) + glob.glob("profiles/r06*bench*.json"))):
    try:
        for line in open(path, errors="ignore"):
            if line.startswith('{"metric"') and '"calibration"' in line:
                d = json.loads(line)
                c = d["calibration"]
                rows.append((path, d["value"], d["ms_per_step"], c["mfma_bf16_tflops"], c["copy_tbps"], d.get("value_normalised"), d["config"]["launch"][:8]))
    except Exception as e:      # noqa: BLE001
        print("skip", path, e, file=sys.stderr)
print(f"{'run':58s} {'frames/s':>9s} {'ms/step':>8s} {'mfma TF/s':>10s} {'copy TB/s':>10s} {'normalised':>11s}  launch")
for r in rows:
    print(f"{r[0][-58:]:58s} {r[1]:9.1f} {r[2]:8.2f} {r[3]:10.1f} {r[4]:10.3f} {r[5] if r[5] is None else round(r[5], 1)!s:>11s}  {r[6]}")
if rows:
    print(f"# medians: mfma {statistics.median(r[3] for r in rows):.1f} TFLOP/s, copy {statistics.median(r[4] for r in rows):.3f} TB/s over {len(rows)} runs")
