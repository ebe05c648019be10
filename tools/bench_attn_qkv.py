#!/usr/bin/env python3
"""QKV-fused window attention forward (stswin_win_attn_qkv_fwd) against the kernel pair it replaces (gathered qkv GEMM + attention
forward) on the stage-1 shape of the training step (M = 65536 token rows, C = 512, 4 heads, 8x8 windows, frame pairs; bf16, HIP
events).  Both definitions of SURVEY 8(d): core flops = 4 NTOK^2 HD per (window, head); projection + core = + 2 M 3C C."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip, ops


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    dev, dt = "cuda", torch.bfloat16
    for name, B, H, W, C, heads, ws, T in (("stage1 B=4 (bench step)", 8, 64, 64, 512, 4, 8, 2), ("stage1 B=8", 16, 64, 64, 512, 4, 8, 2),
                                           ("256x256 B=48 key passes", 96, 32, 32, 512, 4, 8, 2)):
        N, d = ws * ws, C // heads
        nW = (H // ws) * (W // ws)
        nB_ = B * nW
        M = B * T * H * W
        x = torch.randn(M, C, device=dev).to(dt)
        w = (torch.randn(3 * C, C, device=dev) / C ** 0.5).to(dt)
        bq = torch.randn(3 * C, device=dev) * 0.1
        biasT = torch.randn(4, heads, N, N, device=dev)
        bidx = (torch.arange(nW, device=dev) % 4).to(torch.int32)
        rmap = ops.window_rowmap(B, T, H, W, ws, 4, dev)
        qkv = torch.empty(M, 3 * C, dtype=dt, device=dev)
        f_core = 4.0 * (T * N) ** 2 * d * nB_ * heads
        f_proj = 2.0 * M * 3 * C * C

        def pair():
            hip.gemm_nt(x, w, qkv, M=M, a_rows=rmap, bias=bq, scale=d ** -0.5, scale_cols=C)
            return hip.win_attn_fwd(qkv, biasT, None, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C, bias_index=bidx)

        t_g = timeit(lambda: hip.gemm_nt(x, w, qkv, M=M, a_rows=rmap, bias=bq, scale=d ** -0.5, scale_cols=C))
        t_a = timeit(lambda: hip.win_attn_fwd(qkv, biasT, None, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C, bias_index=bidx))
        t_p = timeit(pair)
        t_f = timeit(lambda: hip.win_attn_qkv_fwd(x, rmap, w, bq, biasT, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C, scale=d ** -0.5,
                                                  bias_index=bidx, want_qkv=True))
        t_n = timeit(lambda: hip.win_attn_qkv_fwd(x, rmap, w, bq, biasT, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C, scale=d ** -0.5,
                                                  bias_index=bidx, want_qkv=False))
        print(f"{name}: M = {M}")
        print(f"  qkv GEMM (gather, bias, q scale)      {t_g:8.1f} us  {f_proj / t_g / 1e6:7.1f} TF/s")
        print(f"  attention core (reads q | k | v)      {t_a:8.1f} us  {f_core / t_a / 1e6:7.1f} TF/s   {(4 * M * C * 2) / t_a / 1e6:6.2f} TB/s")
        print(f"  pair, back to back                    {t_p:8.1f} us  {(f_proj + f_core) / t_p / 1e6:7.1f} TF/s = {(f_proj + f_core) / t_p / 1e6 / 25:5.1f} % of the bf16 MFMA peak")
        print(f"  FUSED, q | k | v written for backward {t_f:8.1f} us  {(f_proj + f_core) / t_f / 1e6:7.1f} TF/s = {(f_proj + f_core) / t_f / 1e6 / 25:5.1f} %")
        print(f"  FUSED, no-grad (nothing written)      {t_n:8.1f} us  {(f_proj + f_core) / t_n / 1e6:7.1f} TF/s = {(f_proj + f_core) / t_n / 1e6 / 25:5.1f} %", flush=True)


if __name__ == "__main__":
    main()
