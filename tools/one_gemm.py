import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip
M, N, K = [int(v) for v in sys.argv[1:4]]
kind = sys.argv[4] if len(sys.argv) > 4 else "nt"
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
A = torch.randn(M, K, device="cuda").bfloat16()
if kind == "nt":
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(6):
        hip.gemm_nt(A, W, out, M=M, flags=flags)
else:
    Bt = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.zeros(K, N, device="cuda")
    for _ in range(6):
        hip.gemm_tn(A, Bt, out, Mk=M)
torch.cuda.synchronize()
