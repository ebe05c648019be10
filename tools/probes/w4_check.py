#!/usr/bin/env python3
"""4-wave register-pipelined 256x256 ring (GF_W4R) against the 8-wave ping-pong ring: results per epilogue mode, then times.
Needs a tuning build of the library (STSWIN_TUNING=1 python __graft_entry__.py): the product build ignores the flag (both columns then
run the 8-wave kernel; the `var` column shows which kernel ran: 1 = 8-wave, 14 = 4-wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from stswincl_amd import hip
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)


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


def run(M, N, K, mode, fl, S=1):
    A = (torch.randn(M, K // S if S == 1 else K // S, device=dev) * 0.5).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    bias = torch.randn(N, device=dev) * 0.1
    R = (torch.randn(M, N, device=dev)).to(dt)
    kw = {}
    a_rows = None
    if S > 1:
        a_rows = torch.randint(-1, M, (S, M), device=dev, dtype=torch.int32)
        kw["a_rows"] = a_rows
        kw["S"] = S
    outs = []
    for extra in (0, fl):
        out = torch.zeros(M, N, device=dev, dtype=dt)
        out2 = torch.zeros(M, N, device=dev, dtype=dt) if mode in ("gelu_c2", "gelu_c2d") else None
        cs = torch.zeros(N, device=dev) if mode in ("mulr_cs", "dgelu_cs", "cs") else None
        st = torch.zeros(2, 2 * ((M + 255) // 256), N, device=dev) if mode in ("stats", "bias_stats") else None
        f = extra | hip.GF_BIG
        a = dict(kw)
        if mode in ("bias", "bias_scale", "gelu_c2", "gelu", "bias_resid", "gelu_c2d", "bias_stats"):
            a["bias"] = bias
        if mode == "bias_scale":
            a["scale"], a["scale_cols"] = 0.125, N // 2
        if mode in ("gelu_c2", "gelu", "gelu_c2d"):
            f |= hip.GF_GELU
        if mode == "gelu_c2d":
            f |= hip.GF_C2_DGELU
        if mode in ("bias_resid", "resid"):
            f |= hip.GF_RESID; a["resid"] = R
        if mode == "mulr_cs":
            f |= hip.GF_MUL_R; a["resid"] = R
        if mode == "dgelu_cs":
            f |= hip.GF_MUL_DGELU; a["resid"] = R
        if mode == "relu":
            f |= hip.GF_RELU
        fn = lambda: hip.gemm_nt(A, W, out, M=M, out2=out2, colsum_out=cs, stats_out=st, flags=f, **a)
        fn()
        var = hip.load().stswin_last_variant(0)
        if cs is not None:
            cs0 = cs.clone()
        t = timeit(fn)
        outs.append((out.float(), None if out2 is None else out2.float(), None if cs is None else cs0, None if st is None else st.clone(), t, var))
    (o0, c0, s0, t0, tm0, v0), (o1, c1, s1, t1, tm1, v1) = outs
    err = (o0 - o1).abs().max().item() / max(o0.abs().max().item(), 1e-6)
    msg = f"M={M:6d} N={N:5d} K={K:5d} S={S} {mode:10s} var {v0}->{v1}  out {err:.2e}"
    if c0 is not None:
        msg += f" out2 {(c0 - c1).abs().max().item() / max(c0.abs().max().item(), 1e-6):.2e}"
    if s0 is not None:
        msg += f" colsum {(s0 - s1).abs().max().item() / max(s0.abs().max().item(), 1e-6):.2e}"
    if t0 is not None:
        msg += f" stats {(t0 - t1).abs().max().item() / max(t0.abs().max().item(), 1e-6):.2e}"
    fl_ = 2.0 * M * N * K
    msg += f" | {tm0:7.1f} us {fl_ / tm0 / 1e6:6.0f} TF/s -> {tm1:7.1f} us {fl_ / tm1 / 1e6:6.0f} TF/s  x{tm0 / tm1:.3f}"
    print(msg, flush=True)


def main():
    mode_ = os.environ.get("W4_CHECK")
    fl = {"m32": hip.GF_M32PP, "rs": hip.GF_M32PP | hip.GF_W4R, "rot": hip.GF_ROT, "swp": hip.GF_ROT | hip.GF_W4R}.get(mode_, hip.GF_W4R)
    for mode in ("plain", "bias", "bias_scale", "gelu_c2", "gelu", "bias_resid", "resid", "gelu_c2d", "mulr_cs", "dgelu_cs", "cs", "stats",
                 "bias_stats", "relu"):
        run(4096, 512, 512, mode, fl)
    run(1000, 520, 256, "bias", fl)          # ragged edges
    run(4096, 512, 1152, "plain", fl, S=9)    # gathered rows, tap segments
    for M, N, K in ((65536, 512, 2048), (65536, 2048, 512), (65536, 1536, 512), (65536, 512, 512), (16384, 1024, 4096), (16384, 4096, 1024),
                    (4096, 4096, 4096), (8192, 8192, 8192)):
        run(M, N, K, "bias", fl)
    run(65536, 2048, 512, "gelu_c2d", fl)
    run(65536, 512, 2048, "bias_resid", fl)
    run(65536, 2048, 512, "mulr_cs", fl)
    run(65536, 512, 4608, "stats", fl, S=9)


if __name__ == "__main__":
    main()
