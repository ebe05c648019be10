"""Phase durations inside stswin_conv3x3_c64 (wave 0 of every workgroup, wall-clock stamps at 100 MHz):
    python tools/conv_halo_timeline.py [--dgrad]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stswincl_amd import hip  # noqa: E402


def wgrad(lib, x, dy, f, h, w):
    ts = torch.zeros(256, 16, 4, dtype=torch.int64, device="cuda")
    need = lib.stswin_conv3x3_c64_wgrad_scratch(f, h, w)
    ws = torch.empty(need, dtype=torch.float32, device="cuda")
    for _ in range(3):
        rc = lib.stswin_conv3x3_c64_wgrad(hip._p(dy), hip._p(x), hip._p(ts), 1, 2, hip._p(ws), hip._c_long(ws.numel()), f, h, w, hip._stream())
        assert rc == 0
    torch.cuda.synchronize()
    t = ts.cpu().double() / 100.0
    start = t[:, 15, 0].min()
    print(f"kernel start spread {float(t[:, 15, 0].max() - start):.2f} us; last store issued {float(t[:, 15, 3].max() - start):.2f} us after the first start")
    print(f"pad zeroing, address setup      {float((t[:, 0, 0] - t[:, 15, 0]).mean()):6.2f}")
    for u in range(8):
        d1, d2 = t[:, u, 1] - t[:, u, 0], t[:, u, 2] - t[:, u, 1]
        print(f"unit {u}: copies landed + barrier {float(d1.mean()):6.2f} (p90 {float(d1.quantile(0.9)):5.2f})   fragments + 9 x 16 MFMA {float(d2.mean()):6.2f} (p90 {float(d2.quantile(0.9)):5.2f})")
    print(f"registers -> LDS                {float((t[:, 15, 2] - t[:, 15, 1]).mean()):6.2f}")
    print(f"LDS -> slab                     {float((t[:, 15, 3] - t[:, 15, 2]).mean()):6.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dgrad", action="store_true")
    ap.add_argument("--wgrad", action="store_true")
    a = ap.parse_args()
    f, h, w = 16, 128, 128
    M = f * h * w
    x = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
    res = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
    wt = torch.randn(64, 64, 3, 3, device="cuda") / 24
    ident = torch.arange(64, dtype=torch.int32, device="cuda")
    fwd, dg = hip.conv_pack(wt, torch.bfloat16, ident, ident)
    y = torch.empty_like(x)
    ts = torch.zeros(256, 8, 8, dtype=torch.int64, device="cuda")
    lib = hip.load()
    if a.wgrad:
        return wgrad(lib, x, res, f, h, w)
    for _ in range(3):
        rc = lib.stswin_conv3x3_c64(hip._p(x), hip._p(dg if a.dgrad else fwd), hip._p(y), hip._p(res if a.dgrad else None), hip._p(ts),
                                    f, h, w, -2 if a.dgrad else 2, hip._stream())
        assert rc == 0
    torch.cuda.synchronize()
    t = ts.cpu().double() / 100.0                    # us
    start = t[:, 0, 6].min()
    print(f"kernel start spread over workgroups: {float(t[:, 0, 6].max() - start):.2f} us; last stamp {float(t[:, 3, 5].max() - start):.2f} us after the first")
    print(f"first halo request issue      {float((t[:, 0, 7] - t[:, 0, 6]).mean()):6.2f}   (+ weight loads issue, address setup)")
    names = ["wait halo + stores + barrier", "request next halo", "18 x 16 MFMA", "epilogue (convert, stores)", "statistics"]
    for tile in range(4):
        print(f"tile {tile}:")
        for k, nm in enumerate(names):
            d = t[:, tile, k + 1] - t[:, tile, k]
            print(f"  {nm:30s} {float(d.mean()):6.2f}  p10 {float(d.quantile(0.1)):6.2f}  p90 {float(d.quantile(0.9)):6.2f}")
    print(f"whole workgroup: {float((t[:, 3, 5] - t[:, 0, 6]).mean()):.2f} us")


if __name__ == "__main__":
    main()
