"""Time stswin_conv3x3_c64 against the gather GEMM it replaces on the layer1 shape of the training step
(16 frames x 128 x 128 pixels x 64 channels; forward with the statistics table, input gradient with the residual operand).
    python tools/bench_conv_halo.py [--frames 16] [--hw 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stswincl_amd import hip  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--hw", type=int, default=128)
    a = ap.parse_args()
    f, h, w = a.frames, a.hw, a.hw
    M = f * h * w
    x = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
    wt = torch.randn(64, 64, 3, 3, device="cuda") / 24
    ident = torch.arange(64, dtype=torch.int32, device="cuda")
    fwd, dg = hip.conv_pack(wt, torch.bfloat16, ident, ident)
    fmap = hip.conv_rowmap(f, h, w, h, w, 3, 1, 1, 1, False, "cuda")
    imap = hip.conv_rowmap(f, h, w, h, w, 3, 1, 1, 1, True, "cuda")
    y = torch.empty_like(x)
    tab = hip.stats_table(M, 64, "cuda")
    flops = 2.0 * M * 64 * 576
    byts = 2.0 * M * 64 * 2
    rows = [
        ("fwd  gather GEMM", lambda: hip.gemm_nt(x, fwd, y, M=M, a_rows=fmap, S=9)),
        ("fwd  halo", lambda: hip.conv3x3_c64(x, fwd, y, f, h, w, 1)),
        ("fwd+stats gather GEMM", lambda: hip.gemm_nt(x, fwd, y, M=M, a_rows=fmap, S=9, stats_out=tab)),
        ("fwd+stats halo", lambda: hip.conv3x3_c64(x, fwd, y, f, h, w, 1, stats_out=tab)),
        ("dgrad+resid gather GEMM", lambda: hip.gemm_nt(x, dg, y, M=M, a_rows=imap, S=9, resid=res, flags=hip.GF_RESID)),
        ("dgrad+resid halo", lambda: hip.conv3x3_c64(x, dg, y, f, h, w, -1, resid=res)),
    ]
    dw = torch.empty(64, 576, dtype=torch.float32, device="cuda")
    rows += [
        ("wgrad gemm_tn + combine", lambda: (hip.gemm_tn(res, x, dw, Mk=M, bt_rows=fmap, bseg=64, overwrite=True, tapminor=True), hip.tn_join())),
        ("wgrad halo + fold", lambda: hip.conv3x3_c64_wgrad(res, x, dw, f, h, w, tapminor=True)),
    ]
    for name, fn in rows:
        us = timeit(fn)
        print(f"{name:28s} {us:8.1f} us   {flops / us * 1e-6:7.1f} TFLOP/s   {byts / us * 1e-3:7.1f} GB/s (in + out)")


if __name__ == "__main__":
    main()
