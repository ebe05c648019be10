#!/usr/bin/env python3
"""Randomised parity sweep of gemm_nt / gemm_tn (bf16) against fp32 torch: random shapes (ragged M, N multiples of 8, K multiples
of 64), row maps, epilogue combinations, column sums / statistics tables.  Prints every mismatch; exit code 1 if any."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from stswincl_amd import hip

dev = "cuda"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def check(name, got, ref, tol, info):
    global bad
    err = float((got.float() - ref).abs().max())
    scale = float(ref.abs().max()) + 1e-6
    if not (err <= tol * scale):
        bad += 1
        print(f"MISMATCH {name}: err {err:.4g} scale {scale:.4g} {info}", flush=True)


for case in range(n_cases):
    M = rng.choice([rng.randint(1, 700), rng.randint(700, 9000), rng.randint(9000, 70000), 256 * rng.randint(1, 260)])
    N = 8 * rng.randint(1, 288) if rng.random() < 0.7 else rng.choice([64, 128, 256, 512, 1024, 1536, 2048])
    K = 64 * rng.randint(1, 16)
    S = 9 if rng.random() < 0.25 else 1
    gather = S == 9 or rng.random() < 0.2
    torch.manual_seed(case)
    rows_src = max(M, 64)
    f32 = rng.random() < 0.2                      # the exact-fp32 parity path (128x128 family, 32-wide K tiles)
    DT = torch.float32 if f32 else torch.bfloat16
    TOL = 2e-4 if f32 else 2e-2
    if f32 and M > 20000:
        M = rng.randint(1, 20000)
        rows_src = max(M, 64)
    A = torch.randn(rows_src, K, device=dev).to(DT)
    W = (torch.randn(N, S * K, device=dev) / (S * K) ** 0.5).to(DT)
    a_rows = None
    if gather:
        a_rows = torch.randint(-1 if S == 9 else 0, rows_src, (S, M), device=dev, dtype=torch.int32)
    bias = torch.randn(N, device=dev) if rng.random() < 0.5 else None
    mode = rng.choice(["plain", "relu", "gelu", "gelu_c2d", "resid", "mulr", "colsum", "stats", "scatter"])
    if a_rows is not None:
        Ag = torch.zeros(S, M, K, device=dev)
        for s_ in range(S):
            idx = a_rows[s_].long()
            Ag[s_] = torch.where((idx >= 0)[:, None], A.float()[idx.clamp(min=0)], torch.zeros(1, device=dev))
        lin = torch.einsum("smk,nsk->mn", Ag, W.float().view(N, S, K))
    else:
        lin = A[:M].float() @ W.float().t()
    if bias is not None:
        lin = lin + bias
    # outputs (and residuals) as column slices of wider buffers: pitch != N, first column at a 16-byte (or only 8-byte) offset
    pad_l = rng.choice([0, 0, 8, 16, 4]); pad_r = rng.choice([0, 8, 40])
    obuf = torch.empty(M, pad_l + N + pad_r, device=dev, dtype=DT)
    out = obuf[:, pad_l:pad_l + N]
    info = f"M={M} N={N} K={K} S={S} gather={gather} bias={bias is not None} mode={mode} out_off={pad_l} pitch={obuf.shape[1]} dtype={'f32' if f32 else 'bf16'}"
    kw = dict(M=M, a_rows=a_rows if a_rows is None or S > 1 else a_rows[0].contiguous(), S=S, bias=bias)
    try:
        if mode == "plain":
            hip.gemm_nt(A, W, out, **kw); check(mode, out, lin, TOL, info)
        elif mode == "relu":
            hip.gemm_nt(A, W, out, flags=hip.GF_RELU, **kw); check(mode, out, F.relu(lin), TOL, info)
        elif mode == "gelu":
            hip.gemm_nt(A, W, out, flags=hip.GF_GELU, **kw); check(mode, out, F.gelu(lin), TOL, info)
        elif mode == "gelu_c2d":
            o2 = torch.empty_like(out)
            hip.gemm_nt(A, W, out, out2=o2, flags=hip.GF_GELU | hip.GF_C2_DGELU, **kw)
            x = lin.double()
            d = (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5).float()
            check(mode, out, F.gelu(lin), TOL, info); check(mode + "/d", o2, d, TOL, info)
        elif mode in ("resid", "mulr"):
            R = torch.randn(M, N + pad_r, device=dev).to(DT)[:, :N]
            hip.gemm_nt(A, W, out, resid=R, flags=hip.GF_RESID if mode == "resid" else hip.GF_MUL_R, **kw)
            check(mode, out, lin + R.float() if mode == "resid" else lin * R.float(), TOL, info)
        elif mode == "colsum":
            cs = torch.zeros(N, device=dev)
            hip.gemm_nt(A, W, out, colsum_out=cs, **kw)
            check(mode, out, lin, TOL, info); check(mode + "/cs", cs, lin.sum(0), 5e-3 + 0.02 / (1 + M ** 0.5), info)
        elif mode == "stats":
            if N % 4 or M < 256:
                continue
            tab = hip.stats_table(M, N, dev).fill_(float("nan"))
            hip.gemm_nt(A, W, out, stats_out=tab, **kw)
            s1 = tab[0].sum(0); s2 = tab[1].sum(0)
            check(mode, out, lin, TOL, info); check(mode + "/sum", s1, lin.sum(0), 5e-3 + 0.02 / (1 + M ** 0.5), info)
            check(mode + "/sq", s2, (lin * lin).sum(0), 5e-3, info)
        elif mode == "scatter":
            perm = torch.randperm(M, device=dev).int()
            hip.gemm_nt(A, W, out, c_rows=perm, **kw)
            ref = torch.empty_like(lin); ref[perm.long()] = lin
            check(mode, out, ref, TOL, info)
    except hip.StswinHipError as e:
        print(f"ERROR {e} {info}", flush=True); bad += 1
    # gemm_tn on a related shape
    if case % 3 == 0:
        Mk = rng.choice([rng.randint(64, 5000), 32 * rng.randint(100, 2100)])
        Ni, Nj = 8 * rng.randint(1, 130), 8 * rng.randint(1, 200)
        At = torch.randn(Mk, Ni, device=dev).bfloat16(); Bt = torch.randn(Mk, Nj, device=dev).bfloat16()
        ow = rng.random() < 0.5
        C = torch.empty(Ni, Nj, device=dev) if ow else torch.ones(Ni, Nj, device=dev)
        use_map = rng.random() < 0.3
        rows = torch.randint(0, Mk, (Mk,), device=dev, dtype=torch.int32) if use_map else None
        hip.gemm_tn(At, Bt, C, Mk=Mk, bt_rows=rows, overwrite=ow)
        Bs = Bt.float()[rows.long()] if use_map else Bt.float()
        ref = At.float().t() @ Bs + (0.0 if ow else 1.0)
        check("tn", C, ref, 1e-2, f"Mk={Mk} Ni={Ni} Nj={Nj} overwrite={ow} map={use_map}")
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
