#!/usr/bin/env python3
"""Window-attention kernels on the two stage shapes of the training step (bf16, HIP events): us and TFLOP/s.
FLOPs: forward 4*NTOK^2*HD per (window, head); backward 10*NTOK^2*HD."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip


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
    for name, rows, C, heads, ws, T in (("stage1", 65536, 512, 4, 8, 2), ("stage2", 16384, 1024, 4, 4, 2)):
        N = ws * ws
        ntok = T * N
        nB_ = rows // ntok
        nW = 64
        hd = C // heads
        qkv = (torch.randn(rows, 3 * C, device=dev) * 0.5).to(dt)
        do = torch.randn(rows, C, device=dev).to(dt)
        biasT = torch.randn(heads, N, N, device=dev)
        maskT = torch.zeros(nW, N, N, device=dev)
        dbiasT = torch.zeros(heads, N, N, device=dev)
        cs = torch.zeros(3 * C, device=dev)
        probs = nB_ * heads
        f_fwd, f_bwd = 4.0 * ntok * ntok * hd * probs, 10.0 * ntok * ntok * hd * probs
        # what the shifted blocks of the model pass: bias + mask pre-summed into one slot per distinct window pattern + a window -> slot index
        tab4 = torch.randn(4, heads, N, N, device=dev)
        bidx = (torch.arange(nW, device=dev) % 4).to(torch.int32)
        t = timeit(lambda: hip.win_attn_fwd(qkv, tab4, None, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C, bias_index=bidx))
        print(f"{name} fwd table+index {t:8.1f} us {f_fwd / t / 1e6:7.1f} TF/s")
        t = timeit(lambda: hip.win_attn_bwd(qkv, do, tab4, None, dbiasT, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C,
                                            scale=hd ** -0.5, colsum_out=cs, bias_index=bidx))
        print(f"{name} bwd table+index {t:8.1f} us {f_bwd / t / 1e6:7.1f} TF/s   (the shifted blocks of the training step)", flush=True)
        # BASELINE configs[4]: q | k | v stored as e4m3 + per-(window, head) scales (written by the QKV GEMM epilogue)
        x = torch.randn(rows, C, device=dev).to(dt)
        wq = (torch.randn(3 * C, C, device=dev) / C ** 0.5).to(dt)
        bq = torch.randn(3 * C, device=dev) * 0.1
        q8, sc = hip.gemm_nt_qkv_fp8(x, wq, M=rows, bias=bq, scale=hd ** -0.5, scale_cols=C, rows_per_problem=ntok, head_dim=hd)
        qkv16 = torch.empty(rows, 3 * C, device=dev, dtype=dt)
        t = timeit(lambda: hip.gemm_nt(x, wq, qkv16, M=rows, bias=bq, scale=hd ** -0.5, scale_cols=C))
        t8 = timeit(lambda: hip.gemm_nt_qkv_fp8(x, wq, M=rows, bias=bq, scale=hd ** -0.5, scale_cols=C, rows_per_problem=ntok, head_dim=hd))
        print(f"{name} qkv GEMM  bf16 out {t:8.1f} us   e4m3 out + scales {t8:8.1f} us")
        t = timeit(lambda: hip.win_attn_fwd_f8(q8, sc, tab4, None, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C, bias_index=bidx))
        print(f"{name} fwd fp8-stored q|k|v (table+index) {t:8.1f} us {f_fwd / t / 1e6:7.1f} TF/s")
        t = timeit(lambda: hip.win_attn_bwd_f8(q8, sc, do, tab4, None, dbiasT, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C,
                                               scale=hd ** -0.5, colsum_out=cs, bias_index=bidx))
        print(f"{name} bwd fp8-stored q|k|v (table+index) {t:8.1f} us {f_bwd / t / 1e6:7.1f} TF/s", flush=True)
        for mname, m in (("mask", maskT), ("nomask", None)):
            t = timeit(lambda: hip.win_attn_fwd(qkv, biasT, m, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C))
            print(f"{name} fwd {mname:7s} {t:8.1f} us {f_fwd / t / 1e6:7.1f} TF/s")
            t = timeit(lambda: hip.win_attn_bwd(qkv, do, biasT, m, dbiasT, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C,
                                                scale=hd ** -0.5, colsum_out=cs))
            print(f"{name} bwd {mname:7s} {t:8.1f} us {f_bwd / t / 1e6:7.1f} TF/s", flush=True)
            t = timeit(lambda: hip.win_attn_bwd(qkv, do, biasT, m, None, nB_=nB_, nW=nW, T=T, ws=ws, heads=heads, C=C,
                                                scale=hd ** -0.5, colsum_out=None))
            print(f"{name} bwd {mname:7s} {t:8.1f} us (no dbias / colsum atomics)", flush=True)


if __name__ == "__main__":
    main()
