import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd import hip
torch.manual_seed(5)
for (m, n, k, flags) in [(8292, 512, 128, 0), (8292, 512, 128, hip.GF_NOBIG), (8292, 64, 64, 0), (8488, 768, 64, hip.GF_BIG), (8192, 256, 128, hip.GF_MID)]:
    a = torch.randn(m, k).bfloat16().cuda()
    w = (torch.randn(n, k) / k ** 0.5).bfloat16().cuda()
    out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    cs = torch.zeros(n, device="cuda")
    tab = hip._cs_table(a.device, 1)
    tab.fill_(float("nan"))
    hip.gemm_nt(a, w, out, M=m, flags=flags, colsum_out=cs)
    torch.cuda.synchronize()
    rows = 2 * ((m + 255) // 256)
    t = tab[:rows * n].view(rows, n)
    ref = out.float().sum(0)
    blocks = out.float()[: (m // 128) * 128].view(-1, 128, n).sum(1)
    print(m, n, k, flags, "err", float((cs - ref).abs().max()), "nan rows", torch.isnan(t).any(1).nonzero().flatten().tolist()[:10],
          "nan cols", torch.isnan(t).any(0).nonzero().flatten().tolist()[:10],
          "block err", float((t[: m // 128] - blocks).abs().nan_to_num(1e9).max()))
