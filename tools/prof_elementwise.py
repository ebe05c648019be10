import csv, sys, re
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = max(r[1] for r in rows)
t0 = t_end - int(float(sys.argv[2]) * 1e6)
agg = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    if s < t0 or "at::native" not in n and "rocclr" not in n and "rocprim" not in n:
        continue
    m = re.search(r"(vectorized_elementwise_kernel|elementwise_kernel_manual_unroll|unrolled_elementwise_kernel|reduce_kernel|multi_tensor_apply_kernel|index\w+|CatArray\w+|\w+)[<(]", n)
    fn = re.findall(r"(FillFunctor<\w+>|\w+Functor\w*<[^>]*>|\w*copy_kernel\w*|direct_copy_kernel\w*|\w+_kernel_cuda|gpu_kernel_impl\w*|AUnaryFunctor<[^,]*,[^,]*,[^,]*, at::native::\w+::\w+|BinaryFunctor<[^,]*,[^,]*,[^,]*, at::native::\w+::\w+|threshold\w*|sqrt\w*|where\w*|compare\w*)", n)
    key = (m.group(1) if m else n[:40]) + " :: " + (fn[0][:70] if fn else n[60:130])
    agg[key][0] += 1
    agg[key][1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{c:6d} {t/1e6:8.3f} ms  {k}")
