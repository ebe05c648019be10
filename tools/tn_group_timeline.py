#!/usr/bin/env python3
"""Per-workgroup timeline of the grouped weight-gradient launch (STSWIN_TUNING library: stswin_debug_set_tn_stamps), by problem:
main loop, wait for the tile's other splits, combine - and the same problems launched one by one."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip

dev, dt = "cuda", torch.bfloat16
Mk = 65536
lib = hip.load()
assert hasattr(lib, "stswin_debug_set_tn_stamps"), "needs a STSWIN_TUNING build (STSWIN_HIP_LIB=...)"
A2048 = torch.randn(Mk, 2048, device=dev).to(dt)
A1536 = torch.randn(Mk, 1536, device=dev).to(dt)
A512 = torch.randn(Mk, 512, device=dev).to(dt)
B512 = torch.randn(Mk, 512, device=dev).to(dt)
C512 = torch.randn(Mk, 512, device=dev).to(dt)
D512 = torch.randn(Mk, 512, device=dev).to(dt)
side = int((Mk // 16) ** 0.5)
rmap = hip.win_rowmap(4, 4, side, side, 8, 4)
o1, o2, o3 = torch.empty(2048, 512, device=dev), torch.empty(512, 512, device=dev), torch.empty(1536, 512, device=dev)
plain = os.environ.get("PLAIN") == "1"
probs = [dict(At=A2048, Bt=B512, out=o1, Mk=Mk), dict(At=A512, Bt=C512, out=o2, Mk=Mk, at_rows=None if plain else rmap),
         dict(At=A1536, Bt=D512, out=o3, Mk=Mk, bt_rows=None if plain else rmap)]
stamps = torch.zeros(512, 8, dtype=torch.int64, device=dev)


def report(title):
    torch.cuda.synchronize()
    t = stamps.cpu().double()
    t = t[t[:, 0] > 0]
    t0 = float(t[:, 0].min())
    print(f"{title}: {len(t)} workgroups, span {(float(t[:, 5].max()) - t0) / 100:.1f} us")
    for ni in sorted(set(int(v) for v in t[:, 6])):
        r = t[t[:, 6] == ni]
        d = (r[:, 1:6] - r[:, 0:5]) / 100
        print(f"  Ni = {ni:4d}: {len(r):3d} workgroups  start {float(((r[:, 0] - t0) / 100).mean()):6.1f}  prologue {float(d[:, 0].mean()):5.2f}  first stage {float(d[:, 1].mean()):5.2f}  "
              f"main loop {float(d[:, 2].mean()):6.1f} (max {float(d[:, 2].max()):6.1f})  wait for the tile's splits {float(d[:, 3].mean()):6.1f} (max {float(d[:, 3].max()):6.1f})  "
              f"combine {float(d[:, 4].mean()):5.1f}  end {float(((r[:, 5] - t0) / 100).mean()):6.1f} (max {float(((r[:, 5] - t0) / 100).max()):6.1f})")


lib.stswin_debug_set_tn_stamps(ctypes.c_void_p(stamps.data_ptr()))
for _ in range(3):
    stamps.zero_()
    assert hip.gemm_tn_group(probs)
report(f"grouped, splits {list(hip.LAST_TN_GROUP_SPLITS)}")
for q in probs:
    for _ in range(3):
        stamps.zero_()
        hip.gemm_tn(q["At"], q["Bt"], q["out"], Mk=Mk, at_rows=q.get("at_rows"), bt_rows=q.get("bt_rows"), overwrite=True)
    report("one by one")
lib.stswin_debug_set_tn_stamps(ctypes.c_void_p(0))
