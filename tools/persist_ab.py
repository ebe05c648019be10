#!/usr/bin/env python3
"""Persistent tile loop of the 256x256 ring gemm_nt (tuning builds, STSWIN_NT_PERSIST=1: a grid of 256 workgroups walks the tiles, no
cross-tile overlap) against one workgroup per tile, on the multi-round shapes of the step; back to back and behind a spacer kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
if not hip.tuning_build():
    sys.exit("persist_ab: needs a STSWIN_TUNING build")


def timeit(fn, iters=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


dev, dt = "cuda", torch.bfloat16
G, C2D, RES, MULR = hip.GF_GELU, hip.GF_C2_DGELU, hip.GF_RESID, hip.GF_MUL_R
cases = [("fc1 fwd s1", 65536, 2048, 512, G | C2D, "bias+c2"), ("fc2 dgrad s1", 65536, 2048, 512, MULR, "r+cs"), ("qkv fwd s1", 65536, 1536, 512, 0, "bias"),
         ("plain 65536x2048x512", 65536, 2048, 512, 0, ""), ("proj fwd s1", 65536, 512, 512, RES, "bias+r"), ("fc2 fwd s1", 65536, 512, 2048, RES, "bias+r"),
         ("fc1 fwd s2", 16384, 4096, 1024, G | C2D, "bias+c2"), ("fc1 fwd s1 B=8", 131072, 2048, 512, G | C2D, "bias+c2")]
spacer_buf = torch.randn(65536, 512, device=dev).to(dt)
spacer = lambda: spacer_buf.mul_(1.0)
t_sp = timeit(spacer)
print(f"{'case':24s} {'rounds':>6s}   back to back: tile/WG  persistent     behind a spacer: tile/WG  persistent   (us)")
for name, M, N, K, fl, opts in cases:
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    b = torch.randn(N, device=dev) if "bias" in opts else None
    out2 = torch.empty(M, N, device=dev, dtype=dt) if "c2" in opts else None
    R = torch.randn(M, N, device=dev).to(dt) if "r" in opts.split("+") else None
    cs = torch.zeros(N, device=dev) if "cs" in opts else None
    ref = None
    cells = []
    for v in ("0", "1"):
        os.environ["STSWIN_NT_PERSIST"] = v
        g = lambda: hip.gemm_nt(A, W, out, M=M, bias=b, out2=out2, resid=R, colsum_out=cs, flags=fl)
        g()
        torch.cuda.synchronize()
        if ref is None:
            ref = (out.clone(), out2.clone() if out2 is not None else None)
        else:
            assert torch.equal(out, ref[0]) and (out2 is None or torch.equal(out2, ref[1])), "persistent form changed the result"
        cells.append(timeit(g))

        def both():
            spacer()
            g()
        cells.append(timeit(both) - t_sp)
    print(f"{name:24s} {(M // 256) * (N // 256) / 256:6.1f}   {cells[0]:22.1f} {cells[2]:11.1f}   {cells[1]:24.1f} {cells[3]:11.1f}", flush=True)
os.environ.pop("STSWIN_NT_PERSIST", None)
