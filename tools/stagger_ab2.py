#!/usr/bin/env python3
"""Start-time stagger of the first round of 256x256 ring workgroups (tuning builds, flag bit 17), measured the way a training step sees it:
every GEMM launch is preceded by an HBM-bound spacer kernel (as the LayerNorm in front of fc1), so a launch never inherits the phase
pattern of its predecessor (a back-to-back loop of one GEMM does: the previous launch's staggered tail hides the start delay).
Reported: time of (spacer + GEMM) minus the spacer alone, for no stagger and for several (phases x delay) settings."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
if not hip.tuning_build():
    sys.exit("stagger_ab2: needs a STSWIN_TUNING build")
lib = hip.load()
lib.stswin_debug_set_stagger.restype = ctypes.c_int


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
ST = 1 << 17
cases = [("fc1 fwd gelu+gelu' s1", 65536, 2048, 512, hip.GF_GELU | hip.GF_C2_DGELU, "bias+c2"), ("fc2 dgrad * gelu' + colsum", 65536, 2048, 512, hip.GF_MUL_R, "r+cs"),
         ("qkv fwd", 65536, 1536, 512, 0, "bias"), ("fc1 fwd s2", 16384, 4096, 1024, hip.GF_GELU | hip.GF_C2_DGELU, "bias+c2"),
         ("fc1 fwd nograd", 65536, 2048, 512, hip.GF_GELU, "bias"), ("fc2 fwd + resid", 65536, 512, 2048, hip.GF_RESID, "bias+r")]
settings = [(0, 0), (2, 600), (2, 1200), (4, 300), (4, 500), (8, 150), (8, 250)]
print(f"{'case':28s} " + " ".join(f"{('off' if p == 0 else f'{p}x{t / 100:.1f}us'):>10s}" for p, t in settings) + "   (us per launch behind an HBM-bound spacer)")
spacer_buf = torch.randn(65536, 512, device=dev).to(dt)
for name, M, N, K, fl, opts in cases:
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    b = torch.randn(N, device=dev) if "bias" in opts else None
    out2 = torch.empty(M, N, device=dev, dtype=dt) if "c2" in opts else None
    R = torch.randn(M, N, device=dev).to(dt) if "r" in opts.split("+") else None
    cs = torch.zeros(N, device=dev) if "cs" in opts else None
    spacer = lambda: spacer_buf.mul_(1.0)
    t_sp = timeit(spacer)
    cells = []
    for ph, tk in settings:
        if ph:
            lib.stswin_debug_set_stagger(ph, tk)
        extra = ST if ph else 0

        def both():
            spacer()
            hip.gemm_nt(A, W, out, M=M, bias=b, out2=out2, resid=R, colsum_out=cs, flags=fl | extra)
        cells.append(timeit(both) - t_sp)
    print(f"{name:28s} " + " ".join(f"{c:10.1f}" for c in cells), flush=True)
