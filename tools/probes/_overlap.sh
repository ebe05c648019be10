export TMPDIR=/tmp
rm -rf /tmp/prof_ov
rocprofv3 --output-format csv --kernel-trace -d /tmp/prof_ov -o kt -- python3 bench.py --steps 6 --warmup 3 --size 256 --no-cpu-baseline --no-secondary --no-profile --no-calibration --graph 1 > /tmp/ov.log 2>&1
KT=$(find /tmp/prof_ov -name "*kernel_trace.csv" | head -1)
python3 - "$KT" <<'PY'
import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70], r.get("Queue_Id"), r.get("Stream_Id")))
rows.sort()
rows=rows[len(rows)//2:]          # steady state (replays)
ov=0
qs=set()
for a,b in zip(rows, rows[1:]):
    qs.add((a[3],a[4]))
    if b[0] < a[1]:
        ov+=1
        if ov<=25: print(f"OVERLAP {(a[1]-b[0])/1e3:7.1f} us: [{a[2]}] (q {a[3]} s {a[4]}) still running when [{b[2]}] (q {b[3]} s {b[4]}) starts")
print("kernels", len(rows), "overlapping successors", ov, "queues/streams seen", qs)
PY
