#!/usr/bin/env python3
"""Pixel-contrastive similarity kernels alone (BASELINE configs[3] sizes): the round-1 per-map kernel (two launches, one per
loss direction), the bank kernel in the reference's per-sample mode (one launch for both directions) and against a 65536-entry
inter-video bank, with and without the row-max / log-sum-exp outputs; backward = class sums + dq."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
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
    n, hw, c = 8, 1024, 256
    torch.manual_seed(0)
    q = F.normalize(torch.randn(2 * n * hw, c, device=dev), dim=1).to(dt)
    bank = F.normalize(torch.randn(6, n * hw, c, device=dev), dim=2).to(dt)
    lq = torch.randint(0, 12, (2 * n * hw,), dtype=torch.int32, device=dev)
    lb = torch.randint(0, 12, (6, n * hw), dtype=torch.int32, device=dev)
    gmap = [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]]
    fl = 2.0 * 2 * n * hw * 5 * hw * c
    keys = [bank[i] for i in range(1, 6)]
    lks = [lb[i].view(n, hw) for i in range(1, 6)]
    t = timeit(lambda: (hip.contrast_fwd(q[:n * hw], keys, lq[:n * hw].view(n, hw), lks, n, hw),
                        hip.contrast_fwd(q[n * hw:], keys, lq[n * hw:].view(n, hw), lks, n, hw)))
    print(f"round-1 per-map kernel, 2 launches (both directions): {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s")
    for lse in (False, True):
        t = timeit(lambda: hip.contrast_bank_fwd(q, lq, bank, lb, q_sets=2, q_block=hw, bank_block=hw, gmap=gmap, want_lse=lse))
        print(f"bank kernel, per-sample mode, lse={int(lse)}:            {t:8.1f} us  {fl / t / 1e6:7.1f} TFLOP/s  "
              f"({fl / 2 / c / t / 1e-6:.3e} pairs/s)")
    cnt = torch.ones(2 * n * hw, 5, device=dev)
    dp = torch.randn(2 * n * hw, 5, device=dev)
    t1 = timeit(lambda: hip.contrast_class_sums(bank, lb, hw, 12))
    ks = hip.contrast_class_sums(bank, lb, hw, 12)
    t2 = timeit(lambda: hip.contrast_bank_dq(dp, dp, cnt, lq, ks, q_sets=2, q_block=hw, seg=n * hw, bank_block=hw, gmap=gmap))
    print(f"backward: class sums {t1:.1f} us + dq {t2:.1f} us")
    seg = 65536
    bank = F.normalize(torch.randn(6, seg, c, device=dev), dim=2).to(dt)
    lb = torch.randint(0, 12, (6, seg), dtype=torch.int32, device=dev)
    fl = 2.0 * 2 * n * hw * 5 * seg * c
    for lse, unit in ((False, False), (True, False), (True, True)):
        t = timeit(lambda: hip.contrast_bank_fwd(q, lq, bank, lb, q_sets=2, q_block=n * hw, bank_block=seg, gmap=gmap, inv_tau=10.0,
                                                 want_lse=lse, unit_rows=unit), iters=5)
        print(f"bank kernel, 65536-entry bank x 5 maps, lse={int(lse)}{' (unit rows: fixed reference point)' if unit else ''}:   {t:8.1f} us  "
              f"{fl / t / 1e6:7.1f} TFLOP/s = {fl / t / 1e6 / 2500:.3f} of the bf16 MFMA peak  ({fl / 2 / c / t / 1e-6:.3e} pairs/s)")


if __name__ == "__main__":
    main()
