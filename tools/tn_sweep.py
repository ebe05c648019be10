#!/usr/bin/env python
"""Split-count sweep of gemm_tn on the weight-gradient shapes of the training step (bf16, HIP events).

Rows: Mk Ni Nj bseg map | us for: auto (launcher's choice), the 128x128 kernel at auto / 8 / 16 / 32 splits, and the
256x256 ring forced at ~256/tiles, ~512/tiles splits.  Conv shapes use the real 3x3 tap maps.
"""
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


SHAPES = [  # Mk, Ni, Nj, bseg, kind
    (65536, 512, 4608, 512, "conv"), (65536, 256, 2304, 256, "conv"), (65536, 512, 2304, 256, "conv"),
    (65536, 128, 1152, 128, "conv"), (262144, 64, 576, 64, "conv"), (4096, 512, 9216, 1024, "conv"),
    (65536, 1536, 512, 0, "b"), (16384, 3072, 1024, 0, "b"), (65536, 512, 512, 0, "a"), (16384, 1024, 1024, 0, "a"),
    (65536, 2048, 512, 0, ""), (65536, 512, 2048, 0, ""), (16384, 4096, 1024, 0, ""), (16384, 1024, 4096, 0, ""),
    (1048576, 64, 192, 0, ""),
]


def main():
    dev = "cuda"
    only = sys.argv[1:]
    print(f"{'Mk':>8s} {'Ni':>5s} {'Nj':>5s} {'bseg':>5s} {'map':>4s} | " + " ".join(f"{s:>8s}" for s in ("auto", "t-auto", "t-8", "t-16", "t-32", "t-64", "ring-16", "ring-32", "ring-1r", "ring-2r")))
    for Mk, Ni, Nj, bseg, kind in SHAPES:
        if only and kind not in only and (kind or "plain") not in only:
            continue
        At = torch.randn(Mk, Ni, device=dev).bfloat16()
        at_rows = bt_rows = None
        if kind == "conv":
            side = int((Mk // 16) ** 0.5)
            bt_rows = hip.conv3x3_rowmap(16, side, side, 1)
            Bt = torch.randn(Mk, bseg, device=dev).bfloat16()
        else:
            Bt = torch.randn(Mk, Nj, device=dev).bfloat16()
            side = int((Mk // 16) ** 0.5)
            rmap = hip.win_rowmap(4, 4, side, side, 8, 4)
            if kind == "a":
                at_rows = rmap
            elif kind == "b":
                bt_rows = rmap
        out = torch.zeros(Ni, Nj, device=dev)
        cells = []
        NORING, RING = 1 << 28, 1 << 29
        t256 = ((Ni + 255) // 256) * ((Nj + 255) // 256)
        r1 = max(1, min(256 // t256, Mk // 32 // 8))
        r2 = max(1, min(512 // t256, Mk // 32 // 8))
        for splits in (0, NORING, NORING | 8, NORING | 16, NORING | 32, NORING | 64, RING | 16, RING | 32, RING | r1, RING | r2):
            try:
                t = timeit(lambda: hip.gemm_tn(At, Bt, out, Mk=Mk, at_rows=at_rows, bt_rows=bt_rows, bseg=bseg, splits=splits))
                cells.append(f"{t:8.1f}")
            except Exception:
                cells.append(f"{'err':>8s}")
        print(f"{Mk:8d} {Ni:5d} {Nj:5d} {bseg:5d} {kind:>4s} | " + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
