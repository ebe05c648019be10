#!/bin/bash
# SQ counters of the 256x256 ring gemm_nt on 4096^3 (run on the GPU box through gpurun; separate --pmc passes, kernel trace only).
OUT=${1:-gpurun_out/gemm_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_SALU"; do
  TAG=$(echo $SET | cut -d' ' -f1)
  rm -rf /tmp/pmc_$TAG
  rocprofv3 --output-format csv --pmc $SET --kernel-trace -d /tmp/pmc_$TAG -o p -- python3 $GRAFT_REPO_ROOT/tools/probes/gemm_pmc.py > $GRAFT_REPO_ROOT/$OUT/run_$TAG.log 2>&1
  CC=$(find /tmp/pmc_$TAG -name "*counter_collection.csv" | head -1)
  [ -n "$CC" ] && python3 - "$CC" >> $GRAFT_REPO_ROOT/$OUT/gemm_pmc.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "gemm_nt_ring" in r.get("Kernel_Name", ""):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    v = v[len(v) // 3:]          # skip warm-up launches
    print(f"{k:32s} {sum(v) / len(v):16.0f}  (mean per launch over {len(v)} launches)")
PY
done
cat $GRAFT_REPO_ROOT/$OUT/gemm_pmc.txt
