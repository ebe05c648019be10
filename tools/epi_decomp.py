#!/usr/bin/env python3
"""Where the epilogue time of the 256x256 ring gemm_nt goes on the Swin MLP shapes: each epilogue variant of the training
step timed as is, with its global stores removed (debug bit 20) and with no epilogue at all (debug bit 21)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip

NOSTORE, NOEPI, NOWARM = 1 << 20, 1 << 21, 1 << 18   # (third column: bit 18 = R blocks requested after the main loop instead of in its tail)


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
    dt, dev = torch.bfloat16, "cuda"
    shapes = [(65536, 2048, 512, "fc1 s1"), (16384, 4096, 1024, "fc1 s2"), (65536, 512, 2048, "fc2 s1"), (65536, 512, 512, "proj s1")]
    print(f"{'shape':8s} {'epilogue':14s} {'full':>8s} {'nostore':>8s} {'late R':>8s}   (us; 'none' = no epilogue; third column: R tile requested after the main loop)")
    for M, N, K, note in shapes:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        out = torch.empty(M, N, device=dev, dtype=dt)
        out2 = torch.empty(M, N, device=dev, dtype=dt)
        R = torch.randn(M, N, device=dev).to(dt)
        bias = torch.randn(N, device=dev)
        cs = torch.zeros(N, device=dev)
        cases = [("none", dict(flags=NOEPI)), ("plain", dict()), ("bias", dict(bias=bias)),
                 ("bias+gelu", dict(bias=bias, flags=hip.GF_GELU)),
                 ("gelu+pre", dict(bias=bias, out2=out2, flags=hip.GF_GELU)),
                 ("gelu+dgelu", dict(bias=bias, out2=out2, flags=hip.GF_GELU | hip.GF_C2_DGELU)),
                 ("resid", dict(bias=bias, resid=R, flags=hip.GF_RESID)),
                 ("mul_r", dict(resid=R, flags=hip.GF_MUL_R)),
                 ("mul_r+cs", dict(resid=R, flags=hip.GF_MUL_R, colsum_out=cs)),
                 ("cs", dict(colsum_out=cs)),
                 # the same R loads served by L2 (every row aliases row 0): separates load issue cost from HBM time
                 ("resid L2", dict(bias=bias, resid=R[:1].expand(M, N), flags=hip.GF_RESID)),
                 ("mul_r+cs L2", dict(resid=R[:1].expand(M, N), flags=hip.GF_MUL_R, colsum_out=cs)),
                 # half of the rows alias: half the HBM bytes
                 ]
        for cname, kw in cases:
            cells = []
            for extra in (0, NOSTORE, NOWARM):
                if cname == "none" and extra:
                    continue
                k2 = dict(kw)
                k2["flags"] = k2.get("flags", 0) | extra
                try:
                    cells.append(f"{timeit(lambda: hip.gemm_nt(A, W, out, M=M, **k2)):8.1f}")
                except Exception as e:
                    cells.append(f"{'err':>8s}")
            print(f"{note:8s} {cname:14s} " + " ".join(cells), flush=True)


if __name__ == "__main__":
    main()
