#!/usr/bin/env python3
"""In-kernel timeline of the 256x256 ring GEMM (debug flag 1<<19, STSWIN_TUNING builds only): per workgroup timestamps (100 MHz) at
0 start, 1 prologue issued, 2 first stage landed, 3 main loop done, 4 after barrier, 5 C image written (activation math), 6 C stores
issued, 8 second-output image written, 9 its stores issued, 10 R tile landed.
STSWIN_TL_EPI = plain | bias | gelu | gelu_dgelu | resid | mul_r picks the epilogue (default plain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
if not hip.tuning_build():
    sys.exit("gemm_timeline: the in-kernel stamps exist in STSWIN_TUNING builds only (STSWIN_TUNING=1 python __graft_entry__.py --force)")
TN_MODE = len(sys.argv) > 1 and sys.argv[1] == "tn"
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (65536, 2048, 512)))
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
if os.environ.get("STSWIN_ZERO_DATA") == "1":     # all-zero operands: the matrix pipe draws less power, clocks stay up
    A.zero_(); W.zero_()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
out2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
R = torch.randn(M, N, device="cuda").bfloat16()
bias = torch.randn(N, device="cuda")
EPI = os.environ.get("STSWIN_TL_EPI", "plain")
kw = {"plain": dict(), "bias": dict(bias=bias), "gelu": dict(bias=bias, flags=hip.GF_GELU),
      "gelu_dgelu": dict(bias=bias, out2=out2, flags=hip.GF_GELU | hip.GF_C2_DGELU),
      "resid": dict(bias=bias, resid=R, flags=hip.GF_RESID), "mul_r": dict(resid=R, flags=hip.GF_MUL_R)}[EPI]
base_flags = kw.pop("flags", 0)
VARIANT = int(os.environ.get("STSWIN_TL_VARIANT", str(hip.GF_BIG | hip.GF_NOSTREAM)))   # kernel selection bits (e.g. GF_BIG | GF_DUO)
TILE_M, TILE_N = (int(v) for v in os.environ.get("STSWIN_TL_TILE", "256,256").split(","))
nblk = (M // TILE_M) * (N // TILE_N)
for _ in range(3):
    ts = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
    hip.gemm_nt(A, W, out, M=M, flags=base_flags | VARIANT | (1 << 19) | int(os.environ.get("STSWIN_TL_FLAGS", "0")),
                colsum_out=ts.view(torch.float32), **kw)   # STSWIN_TL_FLAGS: 1048576 = no stores, 2097152 = no epilogue
torch.cuda.synchronize()
t = ts.view(nblk, 16).cpu().double()
t0 = t[:, 0].min()
clk = t[:, 7].clone()
t = (t - t0) / 100.0  # us
last = 9 if EPI == "gelu_dgelu" else 6
print(f"epilogue {EPI}, {M}x{N}x{K}, tile {TILE_M}x{TILE_N}: {nblk} workgroups; kernel span {float(t[:, last].max()):.1f} us; per-workgroup phase durations (us): mean / p10 / p90")
phases = [("prologue issue", 0, 1), ("first stage wait", 1, 2), ("main loop", 2, 3), ("barrier (ring free)", 3, 4)]
if EPI in ("resid", "mul_r"):
    phases += [("R tile wait", 4, 10), ("math + C image", 10, 5)]
else:
    phases += [("math + C image", 4, 5)]
phases += [("C readback + stores", 5, 6)]
if EPI == "gelu_dgelu":
    phases += [("barrier + C2 image", 6, 8), ("C2 readback + stores", 8, 9)]
for n, a_, b_ in phases:
    col = t[:, b_] - t[:, a_]
    print(f"  {n:22s} {float(col.mean()):7.2f} {float(col.quantile(0.1)):7.2f} {float(col.quantile(0.9)):7.2f}")
life = t[:, last] - t[:, 0]
d = torch.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]], 1)
print(f"  workgroup lifetime      {float(life.mean()):7.2f}   (main loop at {float((clk / ((t[:, 3] - t[:, 2]) * 1e3)).median()):.2f} GHz shader clock)")
per_cu = max(1, nblk // 256)
starts = t[:, 0].sort().values
print("  start times of workgroups 0,255,256,511,512 (sorted):", [round(float(starts[i]), 1) for i in (0, 255, 256, 511, 512) if i < nblk])
ends = t[:, last].sort().values
print("  end times (sorted) 255, 511, last:", [round(float(ends[i]), 1) for i in (255, 511, nblk - 1) if i < nblk])
print(f"  gap between a workgroup's end and its successor's start on the same slot ~ {(float(ends[-1]) - per_cu * float(life.mean())) / per_cu:.2f} us per tile")


def tn_timeline(Mk=65536, Ni=2048, Nj=512, mapped=False):
    """Same for the gemm_tn ring kernel (debug: ldc < 0 makes C a timestamp buffer and skips the epilogue)."""
    At = torch.randn(Mk, Ni, device="cuda").bfloat16()
    Bt = torch.randn(Mk, Nj, device="cuda").bfloat16()
    side = int((Mk // 16) ** 0.5)
    rmap = hip.win_rowmap(4, 4, side, side, 8, 4) if mapped else None
    nblk = 512
    for _ in range(3):
        ts = torch.zeros(Ni, Nj, dtype=torch.float32, device="cuda")
        hip.gemm_tn(At, Bt, ts, Mk=Mk, bt_rows=rmap, atomics=True, debug_ts=True, splits=(int(os.environ['STSWIN_TL_SPLITS']) | (1 << 29)) if os.environ.get('STSWIN_TL_SPLITS') else 0)
    torch.cuda.synchronize()
    raw = ts.view(torch.int64).view(-1)[: nblk * 8].view(nblk, 8).cpu().double()
    raw = raw[raw[:, 3] > 0]
    t = raw / 100.0
    d = t[:, 1:4] - t[:, 0:3]
    nst = Mk // 32
    print(f"gemm_tn ring Mk={Mk} Ni={Ni} Nj={Nj} mapped={mapped}: {len(t)} workgroups")
    for i, n in enumerate(["prologue issue", "first stage wait", "main loop"]):
        print(f"  {n:18s} {float(d[:, i].mean()):7.2f} us")
    tiles = ((Ni + 255) // 256) * ((Nj + 255) // 256)
    splits = max(1, len(t) // tiles)
    print(f"  stages per workgroup ~{nst // splits}: {float(d[:, 2].mean()) / (nst / splits):.3f} us per 32-row stage")


if len(sys.argv) > 1 and sys.argv[1] == "tn":
    tn_timeline(mapped=False)
    tn_timeline(Ni=1536, mapped=True)


if os.environ.get("STSWIN_TL_XCD") == "1" and not TN_MODE:
    # per-XCD view (workgroup b runs on XCD b % 8): is the CU-to-CU spread systematic?
    import collections
    print("  per-XCD: mean main loop / mean lifetime / mean end time of the first 256 workgroups (us)")
    for x in range(8):
        idx = torch.arange(x, min(nblk, 256), 8)
        print(f"    XCD {x}: {float(d[idx, 2].mean()):7.2f} {float(life[idx].mean()):7.2f} {float(t[idx, last].mean()):7.2f}   max end {float(t[idx, last].max()):7.2f}")
