"""A/B of two builds of the multi-tensor optimizer kernel (round 6: nontemporal loads / stores of p, g, m, v) on the Adam step of the segmentation
model: stswincl_amd/lib/variants/libstswin_hip_{base,nt}.so, one subprocess per (build, round), alternating; prints min / mean ms per step.
Result of round 6 (profiles/r06_adam_nontemporal_ab.txt): 0.741 -> 0.690 ms, adopted."""
import os, sys, subprocess, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from stswincl_amd.net.Ours.base18 import TswinPlus
    from stswincl_amd.optim import FusedAdam
    torch.manual_seed(0)
    m = TswinPlus(12, (64, 64)).cuda()
    opt = FusedAdam(m.parameters(), 1e-4)
    for p in m.parameters(): p.grad = torch.randn_like(p)
    import stswincl_amd.optim as O
    O._EAGER_REPACK = False      # (time the optimizer kernels alone)
    for _ in range(3): opt.step()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): opt.step()
        b.record(); b.synchronize(); ts.append(a.elapsed_time(b) / 10)
    print("RES", min(ts), sum(ts) / len(ts))
    sys.exit(0)
for r in range(3):
    for name in (("base", "nt") if r % 2 == 0 else ("nt", "base")):
        lib = os.path.join(ROOT, "stswincl_amd", "lib", "variants", f"libstswin_hip_{name}.so")
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, STSWIN_HIP_LIB=lib), capture_output=True, text=True)
        print(name, [l for l in o.stdout.splitlines() if l.startswith("RES")] or o.stderr[-300:], flush=True)
