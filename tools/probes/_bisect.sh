for e in "X=1" "STSWIN_TN_FUSED=0" "STSWIN_NO_NT_SPLITK=1" "STSWIN_NO_ARENA=1" "STSWIN_NO_TN_GROUP=1" "STSWIN_LAZY_REPACK=1"; do
  echo "== $e"; env $e EVERY=1 ROWS=6 timeout 200 python tools/probes/graph_vs_eager.py 12 256 4 2>&1 | grep "^step .*differ\|identical bit\|^    (" | head -4
done
