"""Autograd Functions for the decode head / loss on NHWC token matrices, built from libstswin_hip kernels.

A feature map (F, C, H, W) is handled as a token matrix [M = F*H*W][Cp] (Cp = C padded to the GEMM's K granule where
needed).  Convolutions are the segmented gather GEMM (S = k*k taps), BatchNorm / bilinear / pooling are the HBM-bound
kernels of csrc/headops.hip.  `Layout` describes where logical channels sit inside a padded token matrix.
"""
from __future__ import annotations

import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import hip
from .ops import compute_dtype, _f32

KGRAN = 64  # channel padding granule (bf16 GEMM K tile; also valid for the f32 path)


_INDEX_MAPS = {}
_TORCH_PADVEC = os.environ.get("STSWIN_TORCH_PADVEC") == "1"   # A/B switch: fill + per-segment copies / cat
_BN_KEEP_Y = os.environ.get("STSWIN_BN_KEEP_Y") == "1"      # A/B switch: ReLU mask from the stored output everywhere


class Layout:
    """segments (logical_start, length, padded_start); width = padded channel count."""

    def __init__(self, segs: Sequence[Tuple[int, int, int]], width: int):
        self.segs, self.width = list(segs), width
        self.logical = sum(s[1] for s in self.segs)

    @staticmethod
    def dense(c: int) -> "Layout":
        w = (c + KGRAN - 1) // KGRAN * KGRAN
        return Layout([(0, c, 0)], w)

    @staticmethod
    def concat(parts: Sequence["Layout"]) -> "Layout":
        segs, lo, po = [], 0, 0
        for p in parts:
            segs += [(lo + a, n, po + b) for a, n, b in p.segs]
            lo += p.logical
            po += p.width
        return Layout(segs, po)

    @property
    def is_identity(self) -> bool:
        return len(self.segs) == 1 and self.segs[0] == (0, self.width, 0)

    def pad_vec(self, v: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
        if self.is_identity:
            return v.detach().float().contiguous()
        if v.is_cuda and v.dim() == 1 and not _TORCH_PADVEC:
            return hip.vec_gather(v.detach().float().contiguous(), self.index_map(v.device), fill)
        out = torch.full((self.width,), fill, dtype=torch.float32, device=v.device)
        for a, n, b in self.segs:
            out[b:b + n] = v.detach()[a:a + n]
        return out

    def pad_vecs(self, vs, fills):
        """pad_vec of several vectors (a BatchNorm's weight | bias | running statistics) in ONE launch."""
        vs = [v.detach() for v in vs]
        if (self.is_identity or _TORCH_PADVEC or not all(v.is_cuda and v.dim() == 1 and v.dtype == torch.float32 and v.is_contiguous() for v in vs)
                or len(vs) > 4):
            return [self.pad_vec(v, f) for v, f in zip(vs, fills)]
        return hip.vec_gather_multi(vs, self.index_map(vs[0].device), fills)

    def unpad_vecs(self, vs, into=None):
        """unpad_vec of several vectors in ONE launch; into: tensors that receive the results in place (running statistics)."""
        ok = (not self.is_identity and not _TORCH_PADVEC and len(vs) <= 4
              and all(v.is_cuda and v.dim() == 1 and v.dtype == torch.float32 and v.is_contiguous() for v in vs)
              and (into is None or all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.logical for t in into)))
        if not ok:
            outs = [self.unpad_vec(v) for v in vs]
            if into is not None:
                for t, o in zip(into, outs):
                    t.copy_(o)
                return list(into)
            return outs
        return hip.vec_gather_multi(vs, self.position_map(vs[0].device), None, outs=into)

    def unpad_vec(self, v: torch.Tensor) -> torch.Tensor:
        if self.is_identity:
            return v
        if v.is_cuda and v.dim() == 1 and v.dtype == torch.float32 and not _TORCH_PADVEC:
            return hip.vec_gather(v.contiguous(), self.position_map(v.device), 0.0)
        return torch.cat([v[..., b:b + n] for a, n, b in self.segs], dim=-1)

    def position_map(self, device) -> torch.Tensor:
        """int32 [logical]: padded position of every logical channel (the inverse of index_map), cached like it."""
        key = ("pos", self.key(), str(device))
        m = _INDEX_MAPS.get(key)
        if m is None:
            m = torch.empty(self.logical, dtype=torch.int32)
            for a, n, b in self.segs:
                m[a:a + n] = torch.arange(b, b + n, dtype=torch.int32)
            m = m.to(device)
            _INDEX_MAPS[key] = m
        return m

    def key(self):
        return (tuple(self.segs), self.width)

    def index_map(self, device) -> torch.Tensor:
        """int32 [width]: logical channel of every padded position, -1 for padding.  Cached per (layout, device) in a
        module-level table: Layout objects are created per call (Layout.dense), and an uncached map is a host-to-device
        copy per convolution per step (and illegal during hipGraph capture)."""
        key = (self.key(), str(device))
        m = _INDEX_MAPS.get(key)
        if m is None:
            m = torch.full((self.width,), -1, dtype=torch.int32)
            for a, n, b in self.segs:
                m[b:b + n] = torch.arange(a, a + n, dtype=torch.int32)
            m = m.to(device)
            _INDEX_MAPS[key] = m
        return m


_CW: dict = {}


def _conv_mats(w: torch.Tensor, dt, lin: Layout, lout: Layout, dgrad: bool) -> torch.Tensor:
    """(Cout,Cin,k,k) -> GEMM B matrix.  fwd: [Cout_p][S*Cin_p] (tap-major); dgrad: [Cin_p][S*Cout_p] (the inverse
    row map supplies the geometry, so taps keep their order).  Both come from ONE hip.conv_pack launch per weight
    version (the torch formulation - zero fill, slice copies, cast, twice - was ~6 launches per convolution and step)."""
    key = (id(w), dt, lin.key(), lout.key())
    stamp = (w._version, w.data_ptr(), tuple(w.shape))
    hit = _CW.get(key)
    if hit is None or hit[0]() is not w or hit[1] != stamp:
        omap, imap = lout.index_map(w.device), lin.index_map(w.device)
        fwd, dg = hip.conv_pack(w, dt, omap, imap)
        if len(_CW) > 2048:
            for kk in [kk for kk, v in _CW.items() if v[0]() is None]:
                del _CW[kk]
        hit = (weakref.ref(w), stamp, fwd, dg, omap, imap)       # (the maps are kept for ops.repack's batched re-packing)
        _CW[key] = hit
    return hit[3] if dgrad else hit[2]


_MAPS: dict = {}


def _conv_maps(frames, Hin, Win, k, stride, pad, dil, device):
    """(Hout, Wout, fwd map [k*k][M_out] or None, inverse map [k*k][M_in] or None); cached per geometry."""
    Hout = (Hin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wout = (Win + 2 * pad - dil * (k - 1) - 1) // stride + 1
    if k == 1 and stride == 1 and pad == 0:
        return Hout, Wout, None, None
    key = (frames, Hin, Win, k, stride, pad, dil, str(device))
    hit = _MAPS.get(key)
    if hit is None:
        hit = (hip.conv_rowmap(frames, Hin, Win, Hout, Wout, k, stride, pad, dil, False, device),
               hip.conv_rowmap(frames, Hin, Win, Hout, Wout, k, stride, pad, dil, True, device))
        _MAPS[key] = hit
    return Hout, Wout, hit[0], hit[1]


_RESID_GRAD_LINK = os.environ.get("STSWIN_NO_RESID_GRAD_LINK") != "1"      # (A/B switch)
_HALO_CONV = os.environ.get("STSWIN_NO_HALO_CONV") != "1"                  # (A/B switch)
_HALO_WGRAD = os.environ.get("STSWIN_NO_HALO_WGRAD") != "1"                # (A/B switch)
_STEM_WGRAD = os.environ.get("STSWIN_NO_STEM_WGRAD") != "1"                # (A/B switch)
_STEM_CONV = os.environ.get("STSWIN_NO_STEM_CONV") != "1"                  # (A/B switch)


class ConvTokFn(torch.autograd.Function):
    """k x k convolution (any stride / padding / dilation) over NHWC tokens as segmented gather GEMM: forward gathers
    input pixels per tap, dgrad gathers output-gradient pixels through the inverse tap map, wgrad is ONE transposed GEMM
    whose B operand is tap-segmented.  ASPP.py:13-31, base18.py:60-77, resnet.py:31-38, PixPro_swin_v5.py:24-26."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom, lin, lout, want_stats=False, link=None):
        ctx.link = link
        frames, Hin, Win, k, stride, pad, dil = geom
        dt = compute_dtype(x)
        X = x.detach().to(dt)
        Hout, Wout, fmap, imap = _conv_maps(frames, Hin, Win, k, stride, pad, dil, x.device)
        Mi, Mo = frames * Hin * Win, frames * Hout * Wout
        assert X.shape == (Mi, lin.width) and X.stride(1) == 1 and weight.shape[-1] == k
        y = torch.empty(Mo, lout.width, dtype=dt, device=x.device)
        # want_stats: the GEMM epilogue also leaves per-128-row-block column sums and sums of squares of y (the statistics
        # of the BatchNorm that follows: no colstats pass over y)
        tab = hip.stats_table(Mo, lout.width, x.device) if want_stats else None
        # (dilation >= half the map: every row tile has taps that are padding for all of its rows - the kernel skips them)
        ctx.tapskip = hip.GF_TAPSKIP if (k == 3 and stride == 1 and 2 * dil >= Hin + 3 and Mo <= 32768) else 0
        # (64 -> 64 channels, 3x3 / 1 / 1: the halo kernel - no row map, the input crosses HBM once)
        ctx.halo = (_HALO_CONV and bias is None and lin.is_identity and lout.is_identity and X.is_contiguous()
                    and hip.conv3x3_c64_ok(frames, Hin, Win, lin.width, lout.width, k, stride, pad, dil, dt))
        if ctx.halo:
            hip.conv3x3_c64(X, _conv_mats(weight, dt, lin, lout, False), y, frames, Hin, Win, 1, stats_out=tab)
        else:
            hip.gemm_nt(X, _conv_mats(weight, dt, lin, lout, False), y, M=Mo, a_rows=fmap, S=k * k,
                        bias=lout.pad_vec(bias) if bias is not None else None, stats_out=tab, flags=ctx.tapskip)
        ctx.geom = (frames, Hin, Win)
        ctx.cfg = (k, lin, lout, dt, x.dtype, bias is not None, Mi, Mo)
        ctx.save_for_backward(X, weight, fmap, imap)
        if want_stats:
            ctx.mark_non_differentiable(tab)
            ctx.set_materialize_grads(False)             # (else autograd zero-fills a "gradient" of the table for backward)
            return y, tab
        return y

    @staticmethod
    @hip.tn_deferred_backward
    def backward(ctx, dy, _dtab=None):
        if dy is None:                                    # (grads are not materialised when the statistics table is an output)
            return (None,) * 8
        X, weight, fmap, imap = ctx.saved_tensors
        k, lin, lout, dt, in_dtype, has_bias, Mi, Mo = ctx.cfg
        co, ci = weight.shape[:2]
        S = k * k
        g = dy.detach().to(dt).contiguous()
        # weight gradient first: its split-K combine then runs beside the input-gradient GEMM (hip.tn_deferred)
        # (dense layouts: the GEMM output IS the gradient - 1x1 as it is, k x k with the split-K combine storing [cout][cin][k][k]
        #  (tap-minor) instead of the GEMM's [cout][tap][cin] - written straight into the data-parallel bucket slice)
        from .ops import wgrad_buffer
        dense = lin.is_identity and lout.is_identity
        direct = dense and S == 1
        dwp = (wgrad_buffer(weight, (lout.width, S * lin.width), X.device) if dense
               else torch.empty(lout.width, S * lin.width, dtype=torch.float32, device=X.device))
        if ctx.halo and _HALO_WGRAD and hip.conv3x3_c64_wgrad_ok(*ctx.geom):
            # (x rows through an LDS ring once, dy and x fragments read transposed, taps as address shifts; stores nn.Conv2d's layout)
            hip.conv3x3_c64_wgrad(g, X, dwp, *ctx.geom, tapminor=True)
            tapminor = True
        else:
            hip.gemm_tn(g, X, dwp, Mk=Mo, bt_rows=fmap, bseg=lin.width if fmap is not None else 0, overwrite=True,
                        tapminor=dense and S > 1)
            tapminor = dense and S > 1 and hip.last_tn_tapminor()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(Mi, lin.width, dtype=dt, device=X.device)
            # GradLink: the gradient another consumer of x has already produced (the residual branch of the block, or the other
            # convolution reading x) rides in as the R operand of this GEMM's epilogue instead of an autograd add over the map
            pend = ctx.link.take(dt, dx.shape) if ctx.link is not None else None
            if ctx.halo:
                hip.conv3x3_c64(g, _conv_mats(weight, dt, lin, lout, True), dx, *ctx.geom, -1, resid=pend)
            else:
                hip.gemm_nt(g, _conv_mats(weight, dt, lin, lout, True), dx, M=Mi, a_rows=imap, S=S, resid=pend,
                            flags=(hip.GF_RESID if pend is not None else 0) | ctx.tapskip)
            dx = dx.to(in_dtype)
            if ctx.link is not None and not ctx.link.last():
                ctx.link.put(dx)                          # a later consumer's backward adds it in; autograd gets no gradient from here
                dx = None
        hip.tn_join()                                     # dwp is complete from here on (the layout copies below read it)
        if direct or tapminor:
            dw = dwp                                      # already [co][ci] / [co][ci][k][k] with the strides autograd / DDP buckets expect
        elif dense:
            # (direct-store launch: the result is in GEMM order - and dwp may be the parameter's bucket slice, so permute out of place)
            dw = dwp.clone().view(co, S, ci).permute(0, 2, 1)
        else:
            dw = torch.zeros(co, ci, S, dtype=torch.float32, device=X.device)
            d3 = dwp.view(lout.width, S, lin.width)
            for oa, on, ob in lout.segs:
                for ia, in_, ib in lin.segs:
                    dw[oa:oa + on, ia:ia + in_, :] = d3[ob:ob + on, :, ib:ib + in_].permute(0, 2, 1)
        db = None
        if has_bias:
            dbp = hip.zeros(lout.width, device=X.device)
            hip.colsum(g, dbp)
            db = lout.unpad_vec(dbp)
        if dw is dwp:
            dw = dw.view(co, ci, k, k)
        else:
            # a strided view of GEMM-ordered data: autograd would clone it into the parameter's layout anyway - make that copy land
            # in the parameter's all-reduce bucket slice (dwp itself when the layouts are dense, else asked for now)
            dw = dw.reshape(co, ci, k, k)
            from .dp import grad_dest
            slot = getattr(weight, "_stswin_grad_dest", None)
            if dense and slot is not None and dwp.data_ptr() == slot[1].data_ptr():
                dw = dwp.view(co, ci, k, k).copy_(dw)
            elif not dense and slot is not None:
                dest = grad_dest(weight)
                if dest is not None:
                    dw = dest.copy_(dw)
        return dx, dw, db, None, None, None, None, None


class GradLink:
    """The consumers of ONE tensor inside a residual block (resnet.py:42-51: conv1 and the shortcut - the residual input of bn2,
    or the downsample convolution) pass its gradient along instead of leaving the sum to autograd: every consumer but the last
    to run hands its gradient to the link and returns None, the last one (a convolution: `convs` of them take part) adds the
    pending gradient in its input-gradient GEMM's epilogue and returns the total.  The backward order inside a block is fixed by
    the data flow (bn2 before conv1), between two convolutions it is whatever the engine picks - either works."""

    def __init__(self, convs: int):
        self.left, self.g = convs, None

    def put(self, g):
        self.g = g if self.g is None else self.g + g

    def take(self, dt, shape):
        g, self.g = self.g, None
        self.left -= 1
        if g is None:
            return None
        assert tuple(g.shape) == tuple(shape)
        return g.to(dt).contiguous()

    def last(self):
        return self.left <= 0


_FUSED_BN_STATS = os.environ.get("STSWIN_NO_FUSED_BN_STATS") != "1"      # (A/B switch)


def conv_tokens(x_tok, conv: torch.nn.Conv2d, frames, Hin, Win, lin=None, lout=None, stats=None, link=None):
    """Apply an nn.Conv2d's parameters to a token matrix; returns (y_tokens, Hout, Wout), or with stats = True / False
    (y_tokens, Hout, Wout, table or None): the BatchNorm statistics table of y for batchnorm_tokens(stats=...) when asked for
    (True: a train-mode BatchNorm follows) and the output is eligible (bf16 path, >= 8192 rows in whole 256-row tiles)."""
    k, stride, pad, dil = conv.kernel_size[0], conv.stride[0], conv.padding[0], conv.dilation[0]
    lin = lin or Layout.dense(conv.in_channels)
    lout = lout or Layout.dense(conv.out_channels)
    Hout = (Hin + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Wout = (Win + 2 * pad - dil * (k - 1) - 1) // stride + 1
    Mo = frames * Hout * Wout
    want = (stats and _FUSED_BN_STATS and compute_dtype(x_tok) == torch.bfloat16 and Mo >= 8192 and Mo % 256 == 0
            and lout.width % 4 == 0)
    geom = (frames, Hin, Win, k, stride, pad, dil)
    if want:
        y, tab = ConvTokFn.apply(x_tok, conv.weight, conv.bias, geom, lin, lout, True, link)
    else:
        y, tab = ConvTokFn.apply(x_tok, conv.weight, conv.bias, geom, lin, lout, False, link), None
    return (y, Hout, Wout) if stats is None else (y, Hout, Wout, tab)


_STEM_MAPS: dict = {}


def _stem_rowmap(frames, Ho, Wo, Hs, Ws, device):
    """[4][frames*Ho*Wo] int32: first record of tap row s of every output pixel in the padded space-to-depth image; cached."""
    key = (frames, Ho, Wo, Hs, Ws, str(device))
    hit = _STEM_MAPS.get(key)
    if hit is None:
        f, oy, ox = torch.meshgrid(torch.arange(frames, device=device), torch.arange(Ho, device=device),
                                   torch.arange(Wo, device=device), indexing="ij")
        hit = torch.stack([((f * Hs + oy + s) * Ws + ox).reshape(-1) for s in range(4)]).to(torch.int32).contiguous()
        if len(_STEM_MAPS) > 16:
            _STEM_MAPS.clear()
        _STEM_MAPS[key] = hit
    return hit


def _stem_pack(weight, dt):
    """(64, 3, 7, 7) -> [64][4 tap rows][4 records][16]: W[co][c][2s + dy - 1][2t + dx - 1] at (s, t, (dy*2 + dx)*3 + c), zero elsewhere."""
    w8 = F.pad(weight.detach().float(), (1, 0, 1, 0))
    w8 = w8.view(-1, 3, 4, 2, 4, 2).permute(0, 2, 4, 3, 5, 1).reshape(-1, 4, 4, 12)
    return F.pad(w8, (0, 4)).reshape(-1, 256).to(dt)


def _stem_unpack(dw):
    """the inverse of _stem_pack for the fp32 weight gradient [64][256] -> (64, 3, 7, 7) (a view)."""
    co = dw.shape[0]
    return dw.view(co, 4, 4, 16)[..., :12].reshape(co, 4, 4, 2, 2, 3).permute(0, 5, 1, 3, 2, 4).reshape(co, 3, 8, 8)[:, :, 1:, 1:]


class StemConvFn(torch.autograd.Function):
    """torchvision resnet18.conv1 (7x7 / stride 2 / pad 3, Cin = 3; reference resnet.py:98-102) as a 4 x 4 / stride 1 convolution over
    the 2 x 2 space-to-depth image (hip.stem_s2d): per output pixel four 64-value segments, gathered by the GEMM's row map.  The
    34 MB image replaces the 147-wide patch matrix (403 MB at B = 4 clips of 512x512, 175 us to build) for forward and weight
    gradient alike."""

    @staticmethod
    def forward(ctx, img, weight, dt, want_stats=False):
        F_, _, Hh, Ww = img.shape
        Ho, Wo = (Hh + 6 - 7) // 2 + 1, (Ww + 6 - 7) // 2 + 1
        im = img.detach().float().contiguous()
        A, Hs, Ws = hip.stem_s2d(im, dt)
        rmap = _stem_rowmap(F_, Ho, Wo, Hs, Ws, img.device)
        y = torch.empty(F_ * Ho * Wo, 64, dtype=dt, device=img.device)
        tab = hip.stats_table(y.shape[0], 64, img.device) if want_stats else None    # BatchNorm statistics of y (see ConvTokFn)
        if _STEM_CONV and hip.stem_wgrad_ok(Hh, Ww, dt):   # (same geometry rule: bf16, output rows of whole 128-pixel units)
            hip.stem_conv(A, _stem_pack(weight, dt), y, F_, Hh, Ww, stats_out=tab)
        else:
            hip.gemm_nt(A, _stem_pack(weight, dt), y, M=y.shape[0], a_rows=rmap, S=4, stats_out=tab)
        ctx.dt, ctx.geom = dt, (F_, Hh, Ww)
        ctx.save_for_backward(A, rmap, weight)
        if want_stats:
            ctx.mark_non_differentiable(tab)
            ctx.set_materialize_grads(False)
            return y, tab
        return y

    @staticmethod
    @hip.tn_deferred_backward
    def backward(ctx, dy, _dtab=None):
        if dy is None:
            return None, None, None, None
        A, rmap, weight = ctx.saved_tensors
        dw = torch.empty(64, 256, dtype=torch.float32, device=A.device)
        g = dy.detach().to(ctx.dt).contiguous()
        if _STEM_WGRAD and hip.stem_wgrad_ok(ctx.geom[1], ctx.geom[2], ctx.dt):
            hip.stem_wgrad(g, A, dw, *ctx.geom)            # record rows through an LDS ring once (168 MB instead of 537 MB gathered)
        else:
            hip.gemm_tn(g, A, dw, Mk=rmap.shape[1], bt_rows=rmap, bseg=64, overwrite=True)
            hip.tn_join()
        dwv = _stem_unpack(dw)
        from .dp import grad_dest
        dest = grad_dest(weight) if getattr(weight, "_stswin_grad_dest", None) is not None else None
        return None, (dest.copy_(dwv) if dest is not None else dwv.contiguous()), None, None


def stem_conv_tokens(img, weight, dt, stats=False):
    """-> (tokens [F*Ho*Wo][64], statistics table or None): the stem convolution, with the BatchNorm statistics of its output
    from the GEMM epilogue when asked for and eligible (same rule as conv_tokens)."""
    F_, _, Hh, Ww = img.shape
    M = F_ * ((Hh - 1) // 2 + 1) * ((Ww - 1) // 2 + 1)
    if stats and _FUSED_BN_STATS and dt == torch.bfloat16 and M >= 8192 and M % 256 == 0:
        return StemConvFn.apply(img, weight, dt, True)
    return StemConvFn.apply(img, weight, dt), None


class MaxPoolTokFn(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) on tokens (torchvision resnet18.maxpool)."""

    @staticmethod
    def forward(ctx, x, geom):
        frames, Hh, Ww = geom
        dt = compute_dtype(x)
        X = x.detach().to(dt).contiguous()
        Ho, Wo = (Hh + 2 - 3) // 2 + 1, (Ww + 2 - 3) // 2 + 1
        y = torch.empty(frames * Ho * Wo, X.shape[1], dtype=dt, device=x.device)
        arg = torch.empty(frames * Ho * Wo, X.shape[1], dtype=torch.uint8, device=x.device)
        hip.maxpool3x3s2(X, y, arg, frames, Hh, Ww, Ho, Wo)
        ctx.cfg = (frames, Hh, Ww, Ho, Wo, dt, x.dtype)
        ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        frames, Hh, Ww, Ho, Wo, dt, in_dtype = ctx.cfg
        g = dy.detach().to(dt).contiguous()
        dx = torch.empty(frames * Hh * Ww, g.shape[1], dtype=dt, device=g.device)
        hip.maxpool3x3s2(g, dx, arg, frames, Hh, Ww, Ho, Wo, backward=True)
        return dx.to(in_dtype), None


def combine_bn_stats(mean_r: torch.Tensor, m2_r: torch.Tensor, n_r: torch.Tensor):
    """Chan's parallel combination of per-rank statistics: mean_r, m2_r [R][G][C] (m2 = sum of squared deviations from the
    rank's own mean), n_r [R] rows per group -> (global mean [G][C], biased variance [G][C], total count)."""
    n = n_r.sum()
    w = (n_r / n).view(-1, 1, 1)
    mean = (mean_r * w).sum(0)
    m2 = (m2_r + n_r.view(-1, 1, 1) * (mean_r - mean) ** 2).sum(0)
    return mean, m2 / n, n


def refuse_replica(module) -> None:
    """nn.DataParallel over several GPUs (seg18/train_swin.py:131-135) runs replicas of the model on Python threads of ONE
    process.  The caches of this package (zero arena, packed-weight cache keyed on parameter objects, BatchNorm-counter and
    view-batching contexts) are per process, and DataParallel re-broadcasts the weights and re-packs them every step - racing on
    them silently would be the worst outcome, so a replica refuses to run.  One process per GPU is the supported layout."""
    if getattr(module, "_is_replica", False):
        raise hip.StswinHipError(
            "stswincl_amd modules do not run as nn.DataParallel replicas (several GPUs driven by threads of one process). Use one "
            "process per GPU instead: `python -m torch.distributed.run --nproc-per-node N train.py` with "
            "torch.nn.parallel.DistributedDataParallel or stswincl_amd.dp.GradBucketReducer (see INTEGRATION.md, 'Data parallel'); "
            "nn.DataParallel(model, device_ids=[torch.cuda.current_device()]) - a single device - is fine.")


COLLECTIVES: dict = {}     # collective launches of this module so far, by kind (tests assert the per-step budget)


def _sync_world(bn) -> int:
    import torch.distributed as dist
    if isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def rowmajor(t: torch.Tensor, dt) -> torch.Tensor:
    """t as a [M][C] matrix the kernels can address (unit column stride, 16-byte aligned pitch and base) in dtype dt: a column
    slice of a wider buffer (the gradient of a zero-copy concat part) is passed through as it is, anything else is compacted."""
    t = t.detach()
    if t.dtype != dt:
        t = t.to(dt)
    if (t.dim() == 2 and t.stride(1) == 1 and (t.stride(0) * t.element_size()) % 16 == 0 and t.data_ptr() % 16 == 0
            and t.stride(0) >= t.shape[1]):
        return t
    return t.contiguous()


class ConcatColsFn(torch.autograd.Function):
    """torch.cat(parts, dim=1) of token matrices WITHOUT the copy (ASPP.py:48's 2560-channel concat, base18.py:104's 400-channel
    one): the producers have already written the parts into column slices of `buf` (their `out=` argument), so the forward
    only hands out the buffer and the backward hands every producer its column slice of the gradient as a strided view (the
    backward kernels take a row pitch).  A cat kernel per concat and a split copy per part and step are gone."""

    @staticmethod
    def forward(ctx, buf, widths, *parts):
        off = 0
        for w_, p_ in zip(widths, parts):
            assert p_.data_ptr() == buf.data_ptr() + off * buf.element_size() and p_.shape[1] == w_ and p_.stride(0) == buf.stride(0), \
                "ConcatColsFn: a part does not live in its column slice of the buffer"
            off += w_
        assert off == buf.shape[1]
        ctx.widths = widths
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for w_ in ctx.widths:
            outs.append(g[:, off:off + w_])
            off += w_
        return (None, None, *outs)


def concat_buffer(rows: int, widths, dt, device):
    """-> (buffer [rows][sum widths], column-slice views for the producers' `out=`)."""
    buf = torch.empty(rows, sum(widths), dtype=dt, device=device)
    views, off = [], 0
    for w_ in widths:
        views.append(buf[:, off:off + w_])
        off += w_
    return buf, views


class BNTokFn(torch.autograd.Function):
    """nn.BatchNorm2d (+ residual add + ReLU) on tokens with `groups` independent statistic groups."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, relu, resid, groups, lay, eps, momentum, world=1,
                unit=0, stats=None, link=None, out=None):
        ctx.link = link
        dt = compute_dtype(x)
        X = x.detach().to(dt)
        M, Cp = X.shape
        assert Cp == lay.width
        fused_pad = training and world == 1 and not lay.is_identity      # weight | bias | running statistics padded in ONE launch
        if fused_pad:
            gp, bp, rm_p, rv_p = lay.pad_vecs([gamma, beta, running_mean, running_var], [0.0, 0.0, 0.0, 1.0])
        else:
            gp, bp = lay.pad_vec(gamma), lay.pad_vec(beta)
        rows_total = 0
        if training and world > 1:
            # nn.SyncBatchNorm (PixPro_swin_v5.py:215-228): batch statistics over ALL ranks.  ONE all-gather of the per-group
            # (mean, M2) of every statistic group of this BatchNorm (per-view and per-frame groups of a batched pass included:
            # interleaved groups work like local BatchNorm, so view batching and the clip-major layout stay on under SyncBN)
            # instead of the reference's per-tensor collectives; no host synchronisation: every rank holds the same number of
            # rows per group (DistributedSampler pads the shards to equal length, contrast/data/__init__.py:21-25).
            import torch.distributed as dist
            n_loc = M // groups
            s, ss = hip.colstats(X, groups=groups, unit=unit)
            pivot = (X.view(-1, unit, Cp)[:groups, 0, :] if unit > 0 else X.view(groups, n_loc, Cp)[:, 0, :]).float()
            mean_l = pivot + s / n_loc
            m2_l = ss - s * s / n_loc
            # the rank's rows per group ride in the same payload (third plane): ranks may hold different row counts (an uneven
            # last batch, a sampler without padding) and nn.SyncBatchNorm weighs by count too.  Everything stays on the device.
            pack = torch.stack([mean_l, m2_l, torch.full_like(mean_l, float(n_loc))])   # [3][groups][C]
            allp = torch.empty(world * 3, *pack.shape[1:], dtype=pack.dtype, device=pack.device)   # (gloo wants the dim-0 concat form)
            dist.all_gather_into_tensor(allp, pack)
            COLLECTIVES["syncbn_all_gather"] = COLLECTIVES.get("syncbn_all_gather", 0) + 1
            allp = allp.view(world, 3, *pack.shape[1:])
            n_r = allp[:, 2, 0, 0]                                              # [world] rows per group of every rank
            mean, var, n_tot = combine_bn_stats(allp[:, 0], allp[:, 1], n_r)
            rstd = torch.rsqrt(var + eps)
            unbias = n_tot / (n_tot - 1).clamp(min=1)
            # backward: the kernel divides the cross-rank sums by ITS rows per group; pre-scaling them by n_loc / n_tot makes that 1 / n_tot
            ctx.sum_scale = float(n_loc) / n_tot
            rm, rv = lay.pad_vec(running_mean), lay.pad_vec(running_var, 1.0)
            for g in range(groups):
                rm = (1 - momentum) * rm + momentum * mean[g]
                rv = (1 - momentum) * rv + momentum * var[g] * unbias
            running_mean.copy_(lay.unpad_vec(rm))
            running_var.copy_(lay.unpad_vec(rv))
            mean, rstd = mean.contiguous(), rstd.contiguous()
        elif training:
            # statistics: from the producing convolution's epilogue table when one came along and the groups are whole 256-row
            # tiles (raw sums, no pivot), else a colstats pass over X (pivot-shifted sums)
            raw = stats is not None and groups <= 32 and ((unit % 256 == 0) if unit > 0 else ((M // groups) % 256 == 0))

            def finalize(rm, rv):
                if raw:                                   # table -> mean / rstd / running statistics (one launch up to 8 groups)
                    return hip.bn_table_finalize(stats, M, rm, rv, groups, eps, momentum, unit=unit)
                s, ss = hip.colstats(X, groups=groups, unit=unit)
                return hip.bn_finalize(X, s, ss, rm, rv, groups, eps, momentum, unit=unit)

            if lay.is_identity:
                mean, rstd = finalize(running_mean, running_var)
            else:
                mean, rstd = finalize(rm_p, rv_p)
                lay.unpad_vecs([rm_p, rv_p], into=[running_mean, running_var])      # (one launch, straight into the buffers)
        else:
            mean = lay.pad_vec(running_mean).view(1, Cp).expand(groups, Cp).contiguous()
            rstd = torch.rsqrt(lay.pad_vec(running_var, 1.0) + eps).view(1, Cp).expand(groups, Cp).contiguous()
        if out is not None:                              # a column slice of a concat buffer (ConcatColsFn): written in place
            assert out.shape == (M, Cp) and out.dtype == dt and out.stride(1) == 1
            y = out.view_as(out)
        else:
            y = torch.empty(M, Cp, dtype=dt, device=x.device)
        R = resid.detach().to(dt) if resid is not None else None
        hip.bn_apply(X, mean, rstd, gp, bp, y, resid=R, groups=groups, relu=relu, unit=unit)
        ctx.cfg = (training, relu, groups, lay, dt, x.dtype, resid is not None, world, rows_total)
        ctx.unit = unit
        # without a residual the backward recomputes the ReLU mask from X (one tensor less to read in both of its passes)
        keep_y = relu and (resid is not None or _BN_KEEP_Y)
        ctx.save_for_backward(X, y if keep_y else None, mean, rstd, gp, bp if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        X, y, mean, rstd, gp, bp = ctx.saved_tensors
        training, relu, groups, lay, dt, in_dtype, has_res, world, rows_total = ctx.cfg
        g = rowmajor(dy, dt)
        dx = torch.empty_like(X)
        dres = torch.empty_like(X) if has_res else None
        if training and world > 1:
            import torch.distributed as dist
            s1, s2 = hip.bn_bwd(g, X, y if relu else None, mean, rstd, gp, dx, dres, groups, relu, training, phase=1, beta=bp,
                                unit=ctx.unit)
            loc1, loc2 = s1.clone(), s2.clone()                 # weight/bias grads stay local sums (DDP averages them)
            both = torch.stack([s1, s2])
            dist.all_reduce(both)
            COLLECTIVES["syncbn_all_reduce"] = COLLECTIVES.get("syncbn_all_reduce", 0) + 1
            both = both * ctx.sum_scale                         # (rows of all ranks, a device scalar: see forward)
            hip.bn_bwd(g, X, y if relu else None, mean, rstd, gp, dx, dres, groups, relu, training, phase=2,
                       sums=(both[0].contiguous(), both[1].contiguous()), rows_total=0, beta=bp, unit=ctx.unit)
            s1, s2 = loc1, loc2
        else:
            # (the dx pass also writes s1 | s2 summed over the statistic groups: the parameter gradients, no torch reduction)
            gs = torch.empty(2, X.shape[1], dtype=torch.float32, device=X.device) if groups > 1 else None
            s1, s2 = hip.bn_bwd(g, X, y if relu else None, mean, rstd, gp, dx, dres, groups, relu, training, beta=bp, unit=ctx.unit,
                                group_sums=gs)
            if gs is not None:
                s1, s2, groups = gs[0:1], gs[1:2], 1
        if groups > 1:                                   # s1 / s2 are the two halves of one [2][groups][C] buffer: one reduce
            if s1._base is not None and s2._base is s1._base and s2.storage_offset() == s1.storage_offset() + s1.numel():
                both = torch.as_strided(s1, (2, groups, s1.shape[-1]), (groups * s1.shape[-1], s1.shape[-1], 1)).sum(1)
                s1, s2 = both[0:1], both[1:2]
            else:
                s1, s2 = s1.sum(0, keepdim=True), s2.sum(0, keepdim=True)
        if lay.is_identity:
            dgamma, dbeta = s2[0], s1[0]
        else:
            dgamma, dbeta = lay.unpad_vecs([s2[0].contiguous(), s1[0].contiguous()])
        dres = dres.to(in_dtype) if has_res else None
        if dres is not None and ctx.link is not None:    # the shortcut's gradient travels to conv1's input-gradient GEMM
            ctx.link.put(dres)
            dres = None
        return (dx.to(in_dtype), dgamma, dbeta, None, None, None, None, dres, None, None, None, None, None, None, None, None, None)


class BNTokGroupFn(torch.autograd.Function):
    """Several INDEPENDENT nn.SyncBatchNorm layers (+ ReLU) whose inputs are all known - the five ASPP branches, the three 1x1
    projections of the decode head, conv1 / downsample of a ResNet block - with ONE all-gather for all their batch statistics in the
    forward and ONE all-reduce for all their backward sums (PixPro_swin_v5.py:215-228 converts every BatchNorm of both encoders; the
    reference pays one tiny collective pair per layer, ~90 latency-bound collectives per encoder pass).  Per layer the arithmetic is
    BNTokFn's SyncBatchNorm branch, kernel for kernel: the values are identical to one exchange per layer.
    args: n, then n configuration tuples (relu, groups, lay, eps, momentum, unit, out), then per layer x, gamma, beta, running_mean,
    running_var."""

    @staticmethod
    def forward(ctx, n, *args):
        import torch.distributed as dist
        cfgs, tens = args[:n], args[n:]
        world = dist.get_world_size()
        st, packs = [], []
        for i in range(n):
            relu, groups, lay, eps, momentum, unit, out = cfgs[i]
            x, gamma, beta, rmean, rvar = tens[5 * i:5 * i + 5]
            dt = compute_dtype(x)
            X = x.detach().to(dt)
            M, Cp = X.shape
            assert Cp == lay.width
            n_loc = M // groups
            s, ss = hip.colstats(X, groups=groups, unit=unit)
            pivot = (X.view(-1, unit, Cp)[:groups, 0, :] if unit > 0 else X.view(groups, n_loc, Cp)[:, 0, :]).float()
            mean_l = pivot + s / n_loc
            m2_l = ss - s * s / n_loc
            packs.append(torch.stack([mean_l, m2_l, torch.full_like(mean_l, float(n_loc))]).reshape(3, groups * Cp))
            st.append((X, dt, M, Cp, n_loc, x.dtype))
        pack = torch.cat(packs, dim=1).contiguous()                      # [3][sum of groups_i * C_i]
        allp = torch.empty(world * 3, pack.shape[1], dtype=pack.dtype, device=pack.device)
        dist.all_gather_into_tensor(allp, pack)
        COLLECTIVES["syncbn_all_gather"] = COLLECTIVES.get("syncbn_all_gather", 0) + 1
        allp = allp.view(world, 3, -1)
        outs, saved, meta, off = [], [], [], 0
        for i in range(n):
            relu, groups, lay, eps, momentum, unit, out = cfgs[i]
            x, gamma, beta, rmean, rvar = tens[5 * i:5 * i + 5]
            X, dt, M, Cp, n_loc, in_dtype = st[i]
            blk = allp[:, :, off:off + groups * Cp].reshape(world, 3, groups, Cp)
            off += groups * Cp
            n_r = blk[:, 2, 0, 0]
            mean, var, n_tot = combine_bn_stats(blk[:, 0], blk[:, 1], n_r)
            rstd = torch.rsqrt(var + eps)
            unbias = n_tot / (n_tot - 1).clamp(min=1)
            rm, rv = lay.pad_vec(rmean), lay.pad_vec(rvar, 1.0)
            for g in range(groups):
                rm = (1 - momentum) * rm + momentum * mean[g]
                rv = (1 - momentum) * rv + momentum * var[g] * unbias
            rmean.copy_(lay.unpad_vec(rm))
            rvar.copy_(lay.unpad_vec(rv))
            mean, rstd = mean.contiguous(), rstd.contiguous()
            gp, bp = lay.pad_vec(gamma), lay.pad_vec(beta)
            if out is not None:
                assert out.shape == (M, Cp) and out.dtype == dt and out.stride(1) == 1
                y = out.view_as(out)
            else:
                y = torch.empty(M, Cp, dtype=dt, device=x.device)
            hip.bn_apply(X, mean, rstd, gp, bp, y, resid=None, groups=groups, relu=relu, unit=unit)
            outs.append(y)
            keep_y = relu and _BN_KEEP_Y
            saved += [X, y if keep_y else None, mean, rstd, gp, bp if relu else None]
            meta.append((relu, groups, lay, dt, in_dtype, unit, float(n_loc) / n_tot))
        ctx.meta, ctx.n = meta, n
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        import torch.distributed as dist
        n, sv = ctx.n, ctx.saved_tensors
        work, sums = [], []
        for i in range(n):
            relu, groups, lay, dt, in_dtype, unit, sum_scale = ctx.meta[i]
            X, y, mean, rstd, gp, bp = sv[6 * i:6 * i + 6]
            g = rowmajor(dys[i], dt)
            dx = torch.empty_like(X)
            s1, s2 = hip.bn_bwd(g, X, y if relu else None, mean, rstd, gp, dx, None, groups, relu, True, phase=1, beta=bp, unit=unit)
            work.append((g, dx, s1.clone(), s2.clone()))
            sums.append(torch.stack([s1, s2]).reshape(2, -1))
        both = torch.cat(sums, dim=1).contiguous()
        dist.all_reduce(both)
        COLLECTIVES["syncbn_all_reduce"] = COLLECTIVES.get("syncbn_all_reduce", 0) + 1
        grads, off = [], 0
        for i in range(n):
            relu, groups, lay, dt, in_dtype, unit, sum_scale = ctx.meta[i]
            X, y, mean, rstd, gp, bp = sv[6 * i:6 * i + 6]
            g, dx, loc1, loc2 = work[i]
            Cp = X.shape[1]
            blk = (both[:, off:off + groups * Cp] * sum_scale).reshape(2, groups, Cp)
            off += groups * Cp
            hip.bn_bwd(g, X, y if relu else None, mean, rstd, gp, dx, None, groups, relu, True, phase=2,
                       sums=(blk[0].contiguous(), blk[1].contiguous()), rows_total=0, beta=bp, unit=unit)
            s1, s2 = (loc1.sum(0, keepdim=True), loc2.sum(0, keepdim=True)) if groups > 1 else (loc1, loc2)
            grads += [dx.to(in_dtype), lay.unpad_vec(s2[0]), lay.unpad_vec(s1[0]), None, None]
        return (None,) + (None,) * n + tuple(grads)


class syncbn_group:
    """with syncbn_group() as g: y_i = g.conv_bn_relu(...) for several INDEPENDENT conv -> BatchNorm (-> ReLU) branches; g.results()
    after the block returns their outputs in call order.  Where the BatchNorms are nn.SyncBatchNorm in a process group (the
    contrastive stage), the convolutions run at once and the BatchNorms wait for the end of the block, where BNTokGroupFn normalises
    them all behind ONE statistics all-gather (and one all-reduce in the backward).  Everywhere else every call runs immediately,
    exactly as conv_bn_relu."""

    def __init__(self):
        self.items, self.out = [], []

    def __enter__(self):
        return self

    def conv_bn_relu(self, x_tok, conv, bn, geom, lin=None, lout=None, relu=True, groups=1, out=None):
        training = bn.training or bn.running_mean is None
        if not (training and _sync_world(bn) > 1):
            self.out.append(conv_bn_relu(x_tok, conv, bn, geom, lin=lin, lout=lout, relu=relu, groups=groups, out=out))
            return len(self.out) - 1
        frames, Hh, Ww = geom
        lout = lout or Layout.dense(conv.out_channels)
        y = conv_tokens(x_tok, conv, frames, Hh, Ww, lin, lout, stats=False)[0]
        return self.bn(y, bn, relu=relu, groups=groups, lay=lout, out=out)

    def bn(self, y, bn, relu=True, groups=1, il_frames=0, lay=None, stats=None, out=None):
        """BatchNorm (+ ReLU) of an already computed tensor, arguments as batchnorm_tokens (no residual)."""
        training = bn.training or bn.running_mean is None
        if not (training and _sync_world(bn) > 1):
            self.out.append(batchnorm_tokens(y, bn, relu=relu, groups=groups, lay=lay, il_frames=il_frames, stats=stats, out=out))
            return len(self.out) - 1
        self.out.append(None)
        self.items.append((len(self.out) - 1, y, bn, relu, groups, il_frames, lay or Layout.dense(bn.num_features), out))
        return len(self.out) - 1

    def __exit__(self, et, ev, tb):
        if et is not None or not self.items:
            return False
        cfgs, tens = [], []
        for _, y, bn, relu, groups, il_frames, lay, out in self.items:
            if _BN_VIEWS is not None:                  # view batching: as batchnorm_tokens
                views, clips = _BN_VIEWS
                if il_frames:
                    groups = groups * views
                else:
                    assert groups == 1, "view batching: contiguous multi-group BatchNorm inside an encoder is not supported"
                    groups, il_frames = views, clips
            if bn.num_batches_tracked is not None:
                if _NBT_PENDING is not None:
                    _NBT_PENDING.append((bn.num_batches_tracked, groups))
                else:
                    bn.num_batches_tracked += groups
            unit = (y.shape[0] // il_frames) if (il_frames and groups > 1) else 0
            cfgs.append((relu, groups, lay, bn.eps, bn.momentum if bn.momentum is not None else 0.1, unit, out))
            tens += [y, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        ys = BNTokGroupFn.apply(len(self.items), *cfgs, *tens)
        for (slot, *_), y in zip(self.items, ys):
            self.out[slot] = y
        return False

    def results(self):
        return list(self.out)


_NBT_PENDING = None
_BN_VIEWS = None


class bn_views:
    """with bn_views(V, clips): the batch is `clips` clips = V augmented views x clips/V clips each, ordered view-interleaved
    (clip c of view v at index c * V + v).  Every train-mode BatchNorm inside then keeps separate statistics per view (and
    per frame in the ResNet), and updates its running statistics view by view: numerically the V sequential encoder calls
    of the reference (PixPro_swin_v5.py:331-362), in one pass with V times the rows per kernel."""

    def __init__(self, views: int, clips: int):
        assert clips % views == 0
        self.v = (views, clips) if views > 1 else None

    def __enter__(self):
        global _BN_VIEWS
        self.prev, _BN_VIEWS = _BN_VIEWS, self.v
        return self

    def __exit__(self, *exc):
        global _BN_VIEWS
        _BN_VIEWS = self.prev
        return False



class deferred_bn_counters:
    """Inside this context the num_batches_tracked increments of every batchnorm_tokens call (30 one-element add kernels
    per forward of the segmentation model) are collected and applied by ONE torch._foreach_add_ on exit."""

    def __enter__(self):
        global _NBT_PENDING
        self.prev, _NBT_PENDING = _NBT_PENDING, []
        return self

    def __exit__(self, *exc):
        global _NBT_PENDING
        pend, _NBT_PENDING = _NBT_PENDING, self.prev
        if pend:
            if self.prev is not None:
                self.prev.extend(pend)
            else:
                per_t = {}                                      # a counter met several times (an encoder run twice): one entry
                for t, inc in pend:
                    ent = per_t.setdefault(id(t), [t, 0])
                    ent[1] += inc
                by_inc = {}
                for t, inc in per_t.values():
                    by_inc.setdefault(inc, []).append(t)
                for inc, ts in by_inc.items():
                    torch._foreach_add_(ts, inc)
        return False


def batchnorm_tokens(x, bn: torch.nn.BatchNorm2d, relu=True, resid=None, groups=1, lay: Optional[Layout] = None,
                     il_frames: int = 0, stats=None, resid_link=None, out=None):
    """il_frames = F > 0: the rows are F frames stored clip-major and statistic group g = frames g, g + groups, ... (frame t
    of every clip); 0: `groups` contiguous row blocks."""
    lay = lay or Layout.dense(bn.num_features)
    training = bn.training or bn.running_mean is None
    if _BN_VIEWS is not None and training:
        # several augmented views run as one batch (clips ordered view-interleaved): every view keeps its own batch statistics,
        # exactly as when the encoder is called once per view
        views, clips = _BN_VIEWS
        if il_frames:
            groups = groups * views                        # frame u -> group u % (views * T) = (view, t)
        else:
            assert groups == 1, "view batching: contiguous multi-group BatchNorm inside an encoder is not supported"
            groups, il_frames = views, clips               # clip c -> group c % views
    if training and bn.num_batches_tracked is not None:
        if _NBT_PENDING is not None:
            _NBT_PENDING.append((bn.num_batches_tracked, groups))
        else:
            bn.num_batches_tracked += groups
    world = _sync_world(bn) if training else 1
    return BNTokFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, relu, resid, groups, lay,
                         bn.eps, bn.momentum if bn.momentum is not None else 0.1, world,
                         (x.shape[0] // il_frames) if (il_frames and groups > 1) else 0,
                         stats if (training and world == 1) else None, resid_link if resid is not None else None, out)


class BNReluPoolFn(torch.autograd.Function):
    """nn.BatchNorm2d + ReLU + nn.MaxPool2d(3, 2, 1) on tokens as ONE pass forward (the normalised map is never stored): the tail of
    the torchvision stem, resnet.py:98-102.  Backward: the max-pool scatter (gather form, through the winning-tap bytes) and the two
    BatchNorm passes with the ReLU mask recomputed from the input.  Values, taps and gradients are those of BNTokFn followed by
    MaxPoolTokFn.  (A variant whose BatchNorm passes gathered the pooled gradient themselves - no max-pool backward pass - measured
    SLOWER, 253 us against 68 + 146 us: the gather then runs twice, and it breaks the four-rows-in-flight load batching of those passes.)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, groups, eps, momentum, unit, stats, geom):
        frames, Hh, Ww = geom
        dt = compute_dtype(x)
        X = x.detach().to(dt)
        M, C = X.shape
        if training:
            raw = stats is not None and groups <= 32 and ((unit % 256 == 0) if unit > 0 else ((M // groups) % 256 == 0))
            if raw:
                mean, rstd = hip.bn_table_finalize(stats, M, running_mean, running_var, groups, eps, momentum, unit=unit)
            else:
                s, ss = hip.colstats(X, groups=groups, unit=unit)
                mean, rstd = hip.bn_finalize(X, s, ss, running_mean, running_var, groups, eps, momentum, unit=unit)
        else:
            mean = running_mean.float().view(1, C).expand(groups, C).contiguous()
            rstd = torch.rsqrt(running_var.float() + eps).view(1, C).expand(groups, C).contiguous()
        gp, bp = _f32(gamma), _f32(beta)
        y, arg = hip.bn_relu_pool(X, mean, rstd, gp, bp, frames, Hh, Ww, groups=groups, unit=unit)
        ctx.cfg = (training, groups, unit, geom, x.dtype)
        ctx.save_for_backward(X, arg, mean, rstd, gp, bp)
        return y

    @staticmethod
    def backward(ctx, dy):
        X, arg, mean, rstd, gp, bp = ctx.saved_tensors
        training, groups, unit, (frames, Hh, Ww), in_dtype = ctx.cfg
        g = rowmajor(dy, X.dtype).contiguous()
        Hp, Wp = (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1
        dz = torch.empty_like(X)
        hip.maxpool3x3s2(g, dz, arg, frames, Hh, Ww, Hp, Wp, backward=True)
        dx = torch.empty_like(X)
        gs = torch.empty(2, X.shape[1], dtype=torch.float32, device=X.device) if groups > 1 else None
        s1, s2 = hip.bn_bwd(dz, X, None, mean, rstd, gp, dx, None, groups, True, training, beta=bp, unit=unit, group_sums=gs)
        if gs is not None:
            s1, s2 = gs[0:1], gs[1:2]
        return (dx.to(in_dtype), s2[0], s1[0]) + (None,) * 9


_FUSED_STEM_TAIL = os.environ.get("STSWIN_NO_FUSED_STEM_TAIL") != "1"      # (A/B switch)


def batchnorm_relu_maxpool_tokens(x, bn: torch.nn.BatchNorm2d, geom, groups=1, il_frames: int = 0, stats=None):
    """bn -> relu -> MaxPool2d(3, 2, 1) of a token map [frames*H*W][C] (geom = (frames, H, W)); one fused pass each way where the
    BatchNorm is local (no SyncBatchNorm) - else the two separate operators."""
    frames, Hh, Ww = geom
    training = bn.training or bn.running_mean is None
    world = _sync_world(bn) if training else 1
    if not _FUSED_STEM_TAIL or world > 1 or bn.running_mean is None or x.shape[1] % 8:
        return MaxPoolTokFn.apply(batchnorm_tokens(x, bn, relu=True, groups=groups, il_frames=il_frames, stats=stats), geom)
    if _BN_VIEWS is not None and training:                 # (view batching: as batchnorm_tokens)
        views, clips = _BN_VIEWS
        if il_frames:
            groups = groups * views
        else:
            assert groups == 1, "view batching: contiguous multi-group BatchNorm inside an encoder is not supported"
            groups, il_frames = views, clips
    if training and bn.num_batches_tracked is not None:
        if _NBT_PENDING is not None:
            _NBT_PENDING.append((bn.num_batches_tracked, groups))
        else:
            bn.num_batches_tracked += groups
    unit = (x.shape[0] // il_frames) if (il_frames and groups > 1) else 0
    return BNReluPoolFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, groups, bn.eps,
                              bn.momentum if bn.momentum is not None else 0.1, unit, stats if training else None, geom)


class BilinearTokFn(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=False) on tokens: [F*h*w][C] -> [F*H*W][C]."""

    @staticmethod
    def forward(ctx, x, geom, out=None):
        frames, h, w, H, W = geom
        dt = compute_dtype(x)
        X = x.detach().to(dt).contiguous()
        if out is not None:                              # a column slice of a concat buffer (ConcatColsFn)
            assert out.shape == (frames * H * W, X.shape[1]) and out.dtype == dt and out.stride(1) == 1
            y = out.view_as(out)
        else:
            y = torch.empty(frames * H * W, X.shape[1], dtype=dt, device=x.device)
        hip.bilinear(X, y, frames, h, w, H, W)
        ctx.geom, ctx.dt, ctx.in_dtype = geom, dt, x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        frames, h, w, H, W = ctx.geom
        g = rowmajor(dy, ctx.dt)
        dx = torch.empty(frames * h * w, g.shape[1], dtype=ctx.dt, device=g.device)
        hip.bilinear(g, dx, frames, h, w, H, W, backward=True)
        return dx.to(ctx.in_dtype), None, None


class AvgPoolTokFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d(1) on tokens: [F*HW][C] -> [F][C]   (ASPP.py:43)."""

    @staticmethod
    def forward(ctx, x, frames):
        dt = compute_dtype(x)
        X = x.detach().to(dt).contiguous()
        s, _ = hip.colstats(X, groups=frames, squares=False)
        ctx.cfg = (frames, X.shape[0], dt, x.dtype)
        return (s / (X.shape[0] // frames)).to(dt)

    @staticmethod
    def backward(ctx, dy):
        frames, M, dt, in_dtype = ctx.cfg
        dx = torch.empty(M, dy.shape[1], dtype=dt, device=dy.device)
        hip.rows_broadcast(dy.detach().float().contiguous(), dx, frames, scale=float(frames) / M)
        return dx.to(in_dtype), None


class BroadcastTokFn(torch.autograd.Function):
    """bilinear upsample of a 1x1 map = broadcast: [F][C] -> [F*HW][C]   (ASPP.py:46)."""

    @staticmethod
    def forward(ctx, v, rows_per_frame, out=None):
        dt = compute_dtype(v)
        frames, C = v.shape
        if out is not None:                              # a column slice of a concat buffer (ConcatColsFn)
            assert out.shape == (frames * rows_per_frame, C) and out.dtype == dt and out.stride(1) == 1
            y = out.view_as(out)
        else:
            y = torch.empty(frames * rows_per_frame, C, dtype=dt, device=v.device)
        hip.rows_broadcast(v.detach().float().contiguous(), y, frames)
        ctx.cfg = (frames, dt, v.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        frames, dt, in_dtype = ctx.cfg
        s, _ = hip.colstats(rowmajor(dy, dt), groups=frames, squares=False)
        return s.to(in_dtype), None, None


class LogitsUpFn(torch.autograd.Function):
    """nn.functional.interpolate(output, (H, W), mode='bilinear') of base18.py:106: tokens -> NCHW logits."""

    @staticmethod
    def forward(ctx, tok, geom):
        frames, h, w, H, W, nc = geom
        dt = compute_dtype(tok)
        X = tok.detach().to(dt).contiguous()
        out = torch.empty(frames, nc, H, W, dtype=dt, device=tok.device)
        hip.logits_upsample(X, out, frames, h, w, H, W, nc)
        ctx.geom, ctx.dt, ctx.in_dtype, ctx.width = geom, dt, tok.dtype, X.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        frames, h, w, H, W, nc = ctx.geom
        g = dout.detach().to(ctx.dt).contiguous()
        dtok = torch.zeros(frames * h * w, ctx.width, dtype=ctx.dt, device=g.device)
        hip.logits_upsample(dtok, g, frames, h, w, H, W, nc, backward=True)
        return dtok.to(ctx.in_dtype), None


class OhemCEFn(torch.autograd.Function):
    """OhemCELoss2D (seg18/utils/losses.py:32-40) without the full sort and without a host sync:
    n_hard = #(loss > thresh); if n_hard > n_min: mean of those, else mean of the n_min largest."""

    @staticmethod
    def forward(ctx, logits, labels, n_min, thresh, ignore_index):
        lg = logits.detach()
        if lg.dtype not in (torch.float32, torch.bfloat16):
            lg = lg.float()
        lg = lg.contiguous()
        lab = labels.contiguous()
        loss, stats = hip.ce_fwd(lg, lab, ignore_index, thresh)
        value, sel = hip.ohem_select(loss, stats, int(n_min), float(thresh))
        ctx.ignore_index, ctx.in_dtype = ignore_index, logits.dtype
        ctx.save_for_backward(lg, lab, loss, sel)
        return value

    @staticmethod
    def backward(ctx, g):
        lg, lab, loss, sel = ctx.saved_tensors
        d = hip.ce_bwd(lg, lab, loss, sel, g.detach().float().reshape(1).contiguous(), ctx.ignore_index)
        return d.to(ctx.in_dtype), None, None, None, None


# ---------------------------------------------------------------------------------------------- helpers
def to_tokens(x: torch.Tensor) -> torch.Tensor:
    """(F, C, H, W) logical -> [F*H*W][C] (free when x is channels-last)."""
    f, c, h, w = x.shape
    t = x.permute(0, 2, 3, 1).reshape(f * h * w, c)
    # (a single NCHW frame reshapes to a VIEW with strides (1, H*W): the kernels want rows, found by tests/fuzz/fuzz_ops.py)
    return t if t.is_contiguous() else t.contiguous()


def from_tokens(t: torch.Tensor, f: int, h: int, w: int) -> torch.Tensor:
    """[F*H*W][C] -> logical (F, C, H, W) view with channels-last strides (no copy)."""
    return t.view(f, h, w, t.shape[1]).permute(0, 3, 1, 2)


def pad_cols(t: torch.Tensor, width: int) -> torch.Tensor:
    if t.shape[1] == width:
        return t
    return torch.cat([t, t.new_zeros(t.shape[0], width - t.shape[1])], dim=1)


def conv_bn_relu(x_tok, conv: torch.nn.Conv2d, bn: torch.nn.BatchNorm2d, geom, lin=None, lout=None, relu=True, groups=1,
                 resid=None, out=None):
    """conv -> BatchNorm (-> + resid) (-> ReLU) on tokens; geom = (frames, Hin, Win).  Returns the output tokens (written into
    `out`, a column slice of a concat buffer, when given)."""
    frames, Hh, Ww = geom
    lout = lout or Layout.dense(conv.out_channels)
    training = bn.training or bn.running_mean is None
    y, _, _, tab = conv_tokens(x_tok, conv, frames, Hh, Ww, lin, lout, stats=training and _sync_world(bn) == 1)
    return batchnorm_tokens(y, bn, relu=relu, resid=resid, groups=groups, lay=lout, stats=tab, out=out)


def conv1x1_tokens(x_tok, conv: torch.nn.Conv2d, frames, Hh, Ww, lin=None, lout=None):
    return conv_tokens(x_tok, conv, frames, Hh, Ww, lin, lout)[0]
