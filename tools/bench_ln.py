#!/usr/bin/env python3
"""LayerNorm forward / backward on the shapes of the training step: us and effective TB/s (algorithmic bytes: x + dy read, dx
written, + the old dx when accumulating).  STSWIN_LN_BWD_WAVES=4|8 selects the workgroup size of the backward (A/B)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
from bench_gemm import timeit


def main():
    dev, dt = "cuda", torch.bfloat16
    B = int(os.environ.get("STSWIN_LN_B", "4"))
    shapes = [(B * 16384, 512, 0, 0, 1, "stage 1 norm1 (dxsum)"), (B * 16384, 512, 1, 0, 1, "stage 1 norm2 (dx +=, dxsum)"),
              (B * 4096, 1024, 0, 0, 1, "stage 2 norm1"), (B * 4096, 1024, 1, 0, 1, "stage 2 norm2 (dx +=)"),
              (B * 4096, 2048, 0, 1, 0, "patch merging (4 x 512 gathered)")]
    print(f"# B = {B} clips, waves {os.environ.get('STSWIN_LN_BWD_WAVES', 'auto')}")
    print(f"{'M':>7s} {'C':>5s} | {'fwd us':>8s} {'TB/s':>5s} | {'bwd us':>8s} {'TB/s':>5s}   note")
    for M, C, acc, merge, dxs, note in shapes:
        g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        if merge:
            x = torch.randn(4 * M, C // 4, device=dev).to(dt)
            rows = hip.merge_rowmap(B * 4, 64, 64, device=dev)
            kw = dict(rows=rows, S=4, Cseg=C // 4)
        else:
            x = torch.randn(M, C, device=dev).to(dt)
            kw = {}
        nb = M * C * 2
        nset = max(2, int(1.5e9 // (3 * nb)))          # rotate over > 1 GB of operands: the 256 MB infinity cache must not serve them
        xs = [x] + [torch.randn_like(x) for _ in range(nset - 1)]
        dys = [torch.randn(M, C, device=dev).to(dt) for _ in range(nset)]
        dxs_ = [torch.zeros_like(x) for _ in range(nset)]
        y, mean, rstd = hip.layernorm_fwd(x, g, b, M=M, **kw)
        dg, db = hip.zeros(C, device=dev), hip.zeros(C, device=dev)
        dxs_t = hip.zeros(C, device=dev) if dxs else None
        it = [0]

        def fwd():
            i = it[0] = (it[0] + 1) % nset
            hip.layernorm_fwd(xs[i], g, b, M=M, **kw)

        def bwd():
            i = it[0] = (it[0] + 1) % nset
            hip.layernorm_bwd(dys[i], xs[i], g, mean, rstd, dg, db, M=M, dx=dxs_[i], accumulate=bool(acc), dxsum=dxs_t, **kw)
        t_f = timeit(fwd, iters=40) * 1e3
        t_b = timeit(bwd, iters=40) * 1e3
        print(f"{M:7d} {C:5d} | {t_f:8.1f} {2 * nb / t_f / 1e6:5.2f} | {t_b:8.1f} {(3 + acc) * nb / t_b / 1e6:5.2f}   {note}", flush=True)


if __name__ == "__main__":
    main()
