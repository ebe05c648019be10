#!/usr/bin/env python3
"""Product start-time stagger of the 256x256 ring gemm_nt (STSWIN_NT_STAGGER, read per call) on the ring shapes of the training step,
each launch behind an HBM-bound spacer kernel (a launch in the step never follows itself): off / the launcher's rule forced on for
the shape (min rounds 1, any K) / a few fixed per-phase delays.  Decides the launcher's thresholds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["STSWIN_NT_STAGGER_MIN_ROUNDS"] = "1"      # (read once by the library: the rule then applies to every shape asked for)
os.environ["STSWIN_NT_STAGGER_MAX_NT"] = "1000"
import torch
from stswincl_amd import hip


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
         ("fc1 fwd s1 nograd", 65536, 2048, 512, G, "bias"), ("proj fwd s1", 65536, 512, 512, RES, "bias+r"), ("proj dgrad s1", 65536, 512, 512, 0, "cs"),
         ("fc1 fwd s1 B=2", 32768, 2048, 512, G | C2D, "bias+c2"), ("fc2 dgrad s1 B=2", 32768, 2048, 512, MULR, "r+cs"), ("qkv fwd s1 B=2", 32768, 1536, 512, 0, "bias"),
         ("fc1 fwd s2", 16384, 4096, 1024, G | C2D, "bias+c2"), ("fc2 dgrad s2", 16384, 4096, 1024, MULR, "r+cs"), ("qkv fwd s2", 16384, 3072, 1024, 0, "bias"),
         ("fc2 fwd s1", 65536, 512, 2048, RES, "bias+r"), ("fc1 dgrad s1", 65536, 512, 2048, 0, ""), ("qkv dgrad s1", 65536, 512, 1536, RES, "r"),
         ("fc1 fwd s1 B=8", 131072, 2048, 512, G | C2D, "bias+c2"), ("plain 65536x2048x512", 65536, 2048, 512, 0, "")]
settings = ["0", "1", "100", "150", "200", "250", "350"]
print(f"{'case':24s} {'rounds':>6s} {'nt':>3s} " + " ".join(f"{('off' if v == '0' else 'rule' if v == '1' else v + ' tk'):>8s}" for v in settings) +
      "   (us per launch behind a spacer; tk = 10 ns ticks per phase, 8 phases)")
spacer_buf = torch.randn(65536, 512, device=dev).to(dt)
spacer = lambda: spacer_buf.mul_(1.0)
t_sp = timeit(spacer)
for name, M, N, K, fl, opts in cases:
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    b = torch.randn(N, device=dev) if "bias" in opts else None
    out2 = torch.empty(M, N, device=dev, dtype=dt) if "c2" in opts else None
    R = torch.randn(M, N, device=dev).to(dt) if "r" in opts.split("+") else None
    cs = torch.zeros(N, device=dev) if "cs" in opts else None
    cells = []
    for v in settings:
        os.environ["STSWIN_NT_STAGGER"] = v

        def both():
            spacer()
            hip.gemm_nt(A, W, out, M=M, bias=b, out2=out2, resid=R, colsum_out=cs, flags=fl)
        cells.append(timeit(both) - t_sp)
    rounds = (M // 256) * (N // 256) / 256
    print(f"{name:24s} {rounds:6.1f} {K // 32:3d} " + " ".join(f"{c:8.1f}" for c in cells), flush=True)
os.environ.pop("STSWIN_NT_STAGGER", None)
