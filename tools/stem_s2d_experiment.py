"""Stem convolution (7x7 / 2 / 3, 3 -> 64; resnet.py:98-102) over a 2x2 space-to-depth image instead of materialised 147-wide patches:
four contiguous 64-channel segments (tap rows) per output pixel through the row-map gather of gemm_nt / gemm_tn.  Times both forms."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stswincl_amd import hip  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pack_weight(w):                       # (64, 3, 7, 7) -> [64][4 tap rows][4 tap cols][16]
    w8 = F.pad(w, (1, 0, 1, 0))           # ky' = ky + 1 = 2 s + dy2, kx' = kx + 1 = 2 t + dx2
    w8 = w8.view(64, 3, 4, 2, 4, 2).permute(0, 2, 4, 3, 5, 1).reshape(64, 4, 4, 12)
    return F.pad(w8, (0, 4)).reshape(64, 256)


def s2d_torch(img):                       # [F][3][H][W] fp32 -> [F][H/2 + 3][W/2 + 3][16] bf16 (2 pad records top / left, 1 bottom / right)
    Fr, _, H, W = img.shape
    t = img.view(Fr, 3, H // 2, 2, W // 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(Fr, H // 2, W // 2, 12)
    out = torch.zeros(Fr, H // 2 + 3, W // 2 + 3, 16, dtype=torch.bfloat16, device=img.device)
    out[:, 2:-1, 2:-1, :12] = t.to(torch.bfloat16)
    return out


def main():
    Fr, H, W = 16, 512, 512
    Ho, Wo = H // 2, W // 2
    M = Fr * Ho * Wo
    torch.manual_seed(0)
    img = torch.randn(Fr, 3, H, W, device="cuda")
    w = torch.randn(64, 3, 7, 7, device="cuda") / 12
    s2d = s2d_torch(img)
    Hs, Ws = Ho + 3, Wo + 3
    flat = torch.cat([s2d.view(-1), torch.zeros(64, dtype=torch.bfloat16, device="cuda")])
    A = torch.as_strided(flat, (Fr * Hs * Ws, 64), (16, 1))
    f_, oy, ox = torch.meshgrid(torch.arange(Fr, device="cuda"), torch.arange(Ho, device="cuda"), torch.arange(Wo, device="cuda"), indexing="ij")
    rmap = torch.stack([((f_ * Hs + oy + s) * Ws + ox).reshape(-1) for s in range(4)]).to(torch.int32).contiguous()
    wm = pack_weight(w).to(torch.bfloat16)
    y = torch.empty(M, 64, dtype=torch.bfloat16, device="cuda")
    tab = hip.stats_table(M, 64, "cuda")
    hip.gemm_nt(A, wm, y, M=M, a_rows=rmap, S=4, stats_out=tab)
    ref = F.conv2d(img.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), stride=2, padding=3).permute(0, 2, 3, 1).reshape(M, 64)
    print("fwd max err / max", float((y.float() - ref).abs().max()), float(ref.abs().max()))
    # the current form
    patches = hip.stem_im2col(img, torch.bfloat16, Ho, Wo)
    wm0 = torch.zeros(64, 192, device="cuda")
    wm0[:, :147] = w.permute(0, 2, 3, 1).reshape(64, 147)
    wm0 = wm0.to(torch.bfloat16)
    y0 = torch.empty_like(y)
    print(f"im2col                      {timeit(lambda: hip.stem_im2col(img, torch.bfloat16, Ho, Wo)):8.1f} us")
    print(f"gemm_nt patches (K = 192)   {timeit(lambda: hip.gemm_nt(patches, wm0, y0, M=M, stats_out=tab)):8.1f} us")
    print(f"s2d (torch ops)             {timeit(lambda: s2d_torch(img)):8.1f} us")
    print(f"s2d (hip.stem_s2d)          {timeit(lambda: hip.stem_s2d(img, torch.bfloat16)):8.1f} us")
    from stswincl_amd import headops as Hd
    print(f"weight pack (torch ops)     {timeit(lambda: Hd._stem_pack(w, torch.bfloat16)):8.1f} us")
    A2, _, _ = hip.stem_s2d(img, torch.bfloat16)
    assert torch.equal(A2[:, :16], A[:, :16])
    print(f"gemm_nt s2d gather (K=256)  {timeit(lambda: hip.gemm_nt(A, wm, y, M=M, a_rows=rmap, S=4, stats_out=tab)):8.1f} us")
    y5 = torch.empty_like(y)
    print(f"stem_conv ring kernel       {timeit(lambda: hip.stem_conv(A, wm, y5, Fr, H, W, stats_out=tab)):8.1f} us")
    print(f"stem_conv ring, no stats    {timeit(lambda: hip.stem_conv(A, wm, y5, Fr, H, W)):8.1f} us")
    print("ring vs gather fwd max diff", float((y5.float() - y.float()).abs().max()))
    dy = torch.randn(M, 64, device="cuda").to(torch.bfloat16)
    dw0 = torch.empty(64, 192, dtype=torch.float32, device="cuda")
    dw1 = torch.empty(64, 256, dtype=torch.float32, device="cuda")
    print(f"gemm_tn patches             {timeit(lambda: (hip.gemm_tn(dy, patches, dw0, Mk=M, overwrite=True), hip.tn_join())):8.1f} us")
    print(f"gemm_tn s2d gather          {timeit(lambda: (hip.gemm_tn(dy, A, dw1, Mk=M, bt_rows=rmap, bseg=64, overwrite=True), hip.tn_join())):8.1f} us")
    dw2 = torch.empty(64, 256, dtype=torch.float32, device="cuda")
    print(f"stem_wgrad ring + fold      {timeit(lambda: hip.stem_wgrad(dy, A, dw2, Fr, H, W)):8.1f} us")
    print("ring vs gather max diff", float((dw2 - dw1).abs().max()))
    g0 = dw0[:, :147].reshape(64, 7, 7, 3).permute(0, 3, 1, 2)
    g1 = dw1.view(64, 4, 4, 16)[..., :12].reshape(64, 4, 4, 2, 2, 3).permute(0, 5, 1, 3, 2, 4).reshape(64, 3, 8, 8)[:, :, 1:, 1:]
    print("wgrad max diff / max", float((g0 - g1).abs().max()), float(g0.abs().max()))


if __name__ == "__main__":
    main()
