#!/usr/bin/env python3
"""Would ONE grouped launch of the two late weight gradients of a Swin block (proj: 512 x 512 from gathered rows, 4 tiles; qkv: 1536 x 512,
12 tiles) beat two launches?  A grouped launch = both problems resident together with 16 splits each (64 + 192 workgroups) instead of 64 and
21 splits one after the other.  Emulated with two streams, one workspace each; cold operands (rotated)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stswincl_amd import hip

dev, dt = "cuda", torch.bfloat16
Mk, C = int(os.environ.get("MK", 65536)), int(os.environ.get("CC", 512))
lib = hip.load()
NSET = 6
dx1 = [torch.randn(Mk, C, device=dev).to(dt) for _ in range(NSET)]
o = [torch.randn(Mk, C, device=dev).to(dt) for _ in range(NSET)]
dqkv = [torch.randn(Mk, 3 * C, device=dev).to(dt) for _ in range(NSET)]
xn = [torch.randn(Mk, C, device=dev).to(dt) for _ in range(NSET)]
rmap = torch.randperm(Mk, device=dev).to(torch.int32)
dwp, dwq = torch.empty(C, C, device=dev), torch.empty(3 * C, C, device=dev)
ws1 = torch.empty(64 << 20, dtype=torch.float32, device=dev)
ws2 = torch.empty(64 << 20, dtype=torch.float32, device=dev)
OVER, FORCE = 1 << 27, 1 << 29
p = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)      # noqa: E731


def tn(At, at_rows, Bt, bt_rows, out, splits, ws, stream):
    rc = lib.stswin_gemm_tn(0, p(At), ctypes.c_long(At.stride(0)), p(at_rows), p(Bt), ctypes.c_long(Bt.stride(0)), p(bt_rows), p(out),
                            ctypes.c_long(out.stride(0)), Mk, out.shape[0], out.shape[1], splits, 0, p(ws), ctypes.c_long(ws.numel()),
                            ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0, rc


s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def seq(i, sp_p=OVER, sp_q=OVER):
    tn(dx1[i], rmap, o[i], None, dwp, sp_p, ws1, s0)
    tn(dqkv[i], None, xn[i], rmap, dwq, sp_q, ws2, s0)


def conc(i, sp_p, sp_q):
    e = torch.cuda.Event(); e.record(s0)
    s1.wait_event(e); s2.wait_event(e)
    tn(dx1[i], rmap, o[i], None, dwp, sp_p, ws1, s1)
    tn(dqkv[i], None, xn[i], rmap, dwq, sp_q, ws2, s2)
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    e1.record(s1); e2.record(s2)
    s0.wait_event(e1); s0.wait_event(e2)


def timeit(fn, iters=30):
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


print(f"Mk = {Mk}, C = {C}: proj weight gradient {C} x {C} (gathered A rows) + qkv weight gradient {3 * C} x {C} (gathered B rows)")
print(f"two launches, library's split counts         {timeit(lambda i: seq(i)):7.1f} us")
tp, tq = (256 // ((C // 256) ** 2)), 0
for sp in (8, 16, 32):
    print(f"two launches, {sp:2d} splits each (forced ring)     {timeit(lambda i: seq(i, sp | OVER | FORCE, sp | OVER | FORCE)):7.1f} us")
for sp in (12, 16, 20):
    hip.set_cu_budget(0)
    print(f"two streams at once, {sp:2d} splits each             {timeit(lambda i: conc(i, sp | OVER | FORCE, sp | OVER | FORCE)):7.1f} us")
