#!/usr/bin/env python3
"""In-kernel timeline of the 256x256 ring GEMM (debug flag 1<<19): per workgroup timestamps (100 MHz) at
0 start, 1 prologue issued, 2 first stage landed, 3 main loop done, 4 after barrier, 5 image written, 6 stores issued."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (65536, 2048, 512)))
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
nblk = (M // 256) * (N // 256)
for _ in range(3):
    ts = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
    hip.gemm_nt(A, W, out, M=M, flags=hip.GF_BIG | hip.GF_NOSTREAM | (1 << 19), colsum_out=ts.view(torch.float32))
torch.cuda.synchronize()
t = ts.view(nblk, 8).cpu().double()
t0 = t[:, 0].min()
t = (t - t0) / 100.0  # us
d = t[:, 1:7] - t[:, 0:6]
names = ["prologue issue", "first stage wait", "main loop", "barrier", "epilogue math+image", "readback+stores"]
print(f"{nblk} workgroups; kernel span {float(t[:, 6].max()):.1f} us; per-workgroup phase durations (us): mean / p10 / p90")
for i, n in enumerate(names):
    col = d[:, i]
    print(f"  {n:22s} {float(col.mean()):7.2f} {float(col.quantile(0.1)):7.2f} {float(col.quantile(0.9)):7.2f}")
life = t[:, 6] - t[:, 0]
print(f"  workgroup lifetime      {float(life.mean()):7.2f}")
starts = t[:, 0].sort().values
print("  start times of workgroups 0,255,256,511,512 (sorted):", [round(float(starts[i]), 1) for i in (0, 255, 256, 511, 512) if i < nblk])
ends = t[:, 6].sort().values
print("  end times (sorted) 255, 511:", [round(float(ends[i]), 1) for i in (255, 511) if i < nblk])
