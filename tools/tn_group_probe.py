#!/usr/bin/env python3
"""Where does the grouped weight-gradient launch lose?  Groups of plain / gathered problems against their single launches (cold operands)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip

dev, dt = "cuda", torch.bfloat16
Mk = 65536
NSET = 4
A2048 = [torch.randn(Mk, 2048, device=dev).to(dt) for _ in range(NSET)]
A1536 = [torch.randn(Mk, 1536, device=dev).to(dt) for _ in range(NSET)]
A512 = [torch.randn(Mk, 512, device=dev).to(dt) for _ in range(NSET)]
B512 = [torch.randn(Mk, 512, device=dev).to(dt) for _ in range(NSET)]
C512 = [torch.randn(Mk, 512, device=dev).to(dt) for _ in range(NSET)]
rmap = torch.randperm(Mk, device=dev).to(torch.int32)
o1, o2, o3, o4 = (torch.empty(2048, 512, device=dev), torch.empty(512, 512, device=dev), torch.empty(1536, 512, device=dev),
                  torch.empty(2048, 512, device=dev))


def timeit(fn, iters=24):
    for k in range(4):
        fn(k % NSET)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(iters):
        fn(k % NSET)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def group(i, probs):
    assert hip.gemm_tn_group(probs(i))


def single(i, probs):
    for q in probs(i):
        hip.gemm_tn(q["At"], q["Bt"], q["out"], Mk=Mk, at_rows=q.get("at_rows"), bt_rows=q.get("bt_rows"), overwrite=True)


sets = {
    "fc1 + fc1' (both plain, 16 + 16 tiles)": lambda i: [dict(At=A2048[i], Bt=B512[i], out=o1, Mk=Mk), dict(At=A2048[(i + 1) % NSET], Bt=C512[i], out=o4, Mk=Mk)],
    "fc1 + qkv with a PLAIN B operand (16 + 12 tiles)": lambda i: [dict(At=A2048[i], Bt=B512[i], out=o1, Mk=Mk), dict(At=A1536[i], Bt=C512[i], out=o3, Mk=Mk)],
    "fc1 + qkv with gathered B rows": lambda i: [dict(At=A2048[i], Bt=B512[i], out=o1, Mk=Mk), dict(At=A1536[i], Bt=C512[i], out=o3, Mk=Mk, bt_rows=rmap)],
    "fc1 + proj with gathered A rows (16 + 4 tiles)": lambda i: [dict(At=A2048[i], Bt=B512[i], out=o1, Mk=Mk), dict(At=A512[i], Bt=C512[i], out=o2, Mk=Mk, at_rows=rmap)],
    "fc1 + proj + qkv as in the step": lambda i: [dict(At=A2048[i], Bt=B512[i], out=o1, Mk=Mk), dict(At=A512[i], Bt=C512[i], out=o2, Mk=Mk, at_rows=rmap),
                                                  dict(At=A1536[i], Bt=B512[(i + 1) % NSET], out=o3, Mk=Mk, bt_rows=rmap)],
    "fc1 + proj + qkv, all PLAIN": lambda i: [dict(At=A2048[i], Bt=B512[i], out=o1, Mk=Mk), dict(At=A512[i], Bt=C512[i], out=o2, Mk=Mk),
                                              dict(At=A1536[i], Bt=B512[(i + 1) % NSET], out=o3, Mk=Mk)],
}
if os.environ.get("ONLY_STEP"):
    sets = {k: v for k, v in sets.items() if "as in the step" in k or "gathered B" in k}
for name, probs in sets.items():
    fl = sum(2.0 * Mk * q["out"].shape[0] * q["out"].shape[1] for q in probs(0))
    tg = timeit(lambda i: group(i, probs))
    sp = list(hip.LAST_TN_GROUP_SPLITS)
    ts = timeit(lambda i: single(i, probs))
    print(f"{name:55s} grouped {tg:7.1f} us {fl / tg * 1e-6:7.1f} TF/s (splits {sp})   one by one {ts:7.1f} us {fl / ts * 1e-6:7.1f} TF/s")
