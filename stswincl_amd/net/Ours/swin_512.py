"""MI355X-native drop-in for the reference's ``net.Ours.swin_512`` (seg18/net/Ours/swin_512.py; the
contrastive package's ``Ours/swin_tem.py`` and segcata's ``swin_tem_cata.py`` are the same module).

Same classes, constructor signatures, parameter / buffer names and shapes (so reference checkpoints load),
same ``forward`` contracts; the arithmetic runs in libstswin_hip (stswincl_amd.ops).  Tensors must live on
the GPU: there is no CPU path.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from ... import ops


def to_2tuple(v):
    return v if isinstance(v, tuple) else (v, v)


def _relative_position_index(ws_h: int, ws_w: int) -> torch.Tensor:
    """(Wh*Ww, Wh*Ww) index into the (2Wh-1)(2Ww-1) bias table; swin_512.py:89-99."""
    ch, cw = torch.meshgrid(torch.arange(ws_h), torch.arange(ws_w), indexing="ij")
    flat = torch.stack([ch.reshape(-1), cw.reshape(-1)])
    rel = flat[:, :, None] - flat[:, None, :]
    return (rel[0] + ws_h - 1) * (2 * ws_w - 1) + (rel[1] + ws_w - 1)


def window_partition(x: torch.Tensor, window_size: int) -> torch.Tensor:
    """(B, H, W, C) -> (num_windows*B, ws, ws, C) on the GPU (bit-exact copy kernel); swin_512.py:26-38."""
    B, H, W, C = x.shape
    rows = ops.hip.win_move(x.reshape(B, 1, H * W, C), B, 1, H, W, window_size, 0, 0)
    return rows.view(-1, window_size, window_size, C)


def window_reverse(windows: torch.Tensor, window_size: int, H: int, W: int, T: int) -> torch.Tensor:
    """(B*nW, T, ws*ws, C) -> (B, T, H, W, C); swin_512.py:57-71."""
    C = windows.shape[-1]
    B = windows.shape[0] // ((H // window_size) * (W // window_size))
    out = ops.hip.win_move(windows.reshape(-1, C), B, T, H, W, window_size, 0, 1)
    return out.view(B, T, H, W, C)


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2; swin_512.py:7-23 (drop must be 0: the reference never sets it)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        assert drop == 0. and act_layer is nn.GELU
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        x = ops.LinearFn.apply(x, self.fc1.weight, self.fc1.bias, "gelu")
        return ops.LinearFn.apply(x, self.fc2.weight, self.fc2.bias, None)


class WindowAttention(nn.Module):
    """Temporal window attention with tiled relative-position bias; swin_512.py:73-141."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        assert attn_drop == 0. and proj_drop == 0. and qk_scale is None and window_size[0] == window_size[1]
        self.dim = dim
        self.window_size = window_size
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(*window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02, a=-2., b=2.)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x_v, mask=None):
        """x_v (num_windows*B, T, N, C), mask (num_windows, N, N) or None -> (num_windows*B, T, N, C)."""
        return ops.WindowAttentionFn.apply(x_v, self.qkv.weight, self.qkv.bias, self.relative_position_bias_table,
                                           self.proj.weight, self.proj.bias, self.relative_position_index, mask,
                                           self.window_size[0], self.num_heads)


def _shift_mask(H, W, ws, shift):
    """SW-MSA mask (nW, ws*ws, ws*ws) in {0,-100}; swin_512.py:171-190."""
    img = torch.zeros(H, W)
    region = 0
    for h0, h1 in ((0, H - ws), (H - ws, H - shift), (H - shift, H)):
        for w0, w1 in ((0, W - ws), (W - ws, W - shift), (W - shift, W)):
            img[h0:h1, w0:w1] = region
            region += 1
    mw = img.reshape(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


class SwinTransformerBlock(nn.Module):
    """(B, 2, L, C) -> (B, 2, L, C), post-norm topology of swin_512.py:196-237, one fused HIP pipeline."""

    def __init__(self, dim, input_resolution, num_heads, window_size=8, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        assert drop == 0. and attn_drop == 0. and drop_path == 0.
        self.dim = dim
        self.input_resolution = tuple(input_resolution)
        self.num_heads = num_heads
        self.window_size = window_size
        self.shift_size = shift_size
        self.mlp_ratio = mlp_ratio
        if min(self.input_resolution) <= self.window_size:
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=to_2tuple(self.window_size), num_heads=num_heads,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        H, W = self.input_resolution
        attn_mask = _shift_mask(H, W, self.window_size, self.shift_size) if self.shift_size > 0 else None
        self.register_buffer("attn_mask", attn_mask)

    def forward(self, x_v, src=None, out=None):
        """x_v (B, 2, L, C) -> (B, 2, L, C).  Extensions used by SwinTransformerLayerv5's zero-copy schedule: `src` (ops.PairSource):
        x_v is a token matrix [rows][C] and src.xmap locates the frame pairs in it; `out`: [B*2*L][C] destination of the result."""
        H, W = self.input_resolution
        if src is None:
            B, T, L, C = x_v.shape
            assert L == H * W, "input feature has wrong size"
            assert T == 2, "input feature has wrong size"
        a, m = self.attn, self.mlp
        return ops.SwinBlockFn.apply(
            x_v, a.qkv.weight, a.qkv.bias, a.relative_position_bias_table, a.proj.weight, a.proj.bias,
            self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias,
            m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, a.relative_position_index, self.attn_mask,
            (H, W, self.window_size, self.shift_size, self.num_heads), src, out)


class PatchMerging(nn.Module):
    """(B, 4, L, C) -> (B, 4, L/4, 2C); swin_512.py:239-277."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution = tuple(input_resolution)
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        return ops.PatchMergeFn.apply(x, self.norm.weight, self.norm.bias, self.reduction.weight, self.input_resolution)


_FRAME_GRAD_LINK = os.environ.get("STSWIN_NO_FRAME_GRAD_LINK") != "1"      # (A/B switch)
_ZERO_COPY_PAIRS = os.environ.get("STSWIN_NO_ZERO_COPY_PAIRS") != "1"      # (A/B switch: the slice / cat schedule of rounds 1-2)


class _TakeFramesFn(torch.autograd.Function):
    """x_v[:, start:stop] as a contiguous tensor - the input of the layer that runs on the middle frame pair
    (swin_512.py:302-307).  Together with _PutFramesFn it hands x_v ONE gradient tensor: autograd on slice + cat builds three
    zero-filled full-size tensors and adds them up (6 passes over the clip per layer); here the gradient of the output is
    passed through as the gradient of x_v, and the gradient that comes back out of the layer overwrites its frames
    start:stop in place (they are dead by then: only the layer's backward read them)."""

    @staticmethod
    def forward(ctx, x_v, start, stop, link):
        ctx.sl, ctx.link, ctx.shape = (start, stop), link, x_v.shape
        return x_v[:, start:stop].contiguous()

    @staticmethod
    def backward(ctx, g):
        start, stop = ctx.sl
        full = ctx.link.pop("g", None)
        if full is None:                     # the output of _PutFramesFn got no gradient: this is x_v's only one
            full = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
            full[:, start:stop].copy_(g)
            return full, None, None, None
        full[:, start:stop].copy_(g)         # `full` already travels to x_v's producer as _PutFramesFn's gradient
        return None, None, None, None


class _FenceFn(torch.autograd.Function):
    """Identity whose output is consumed by _TakeFramesFn and _PutFramesFn only: the gradient buffer they share is then never
    summed with a third consumer's gradient before _TakeFramesFn has written into it."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _PutFramesFn(torch.autograd.Function):
    """x_v with frames start:stop replaced by `mid` (the other frames pass through)."""

    @staticmethod
    def forward(ctx, x_v, mid, start, stop, link):
        ctx.sl, ctx.link = (start, stop), link
        return torch.cat([x_v[:, :start], mid, x_v[:, stop:]], dim=1)

    @staticmethod
    def backward(ctx, g):
        start, stop = ctx.sl
        if not g.is_contiguous() or g._base is not None:      # never write into somebody else's storage
            g = g.contiguous() if not g.is_contiguous() else g.clone()
        gm = g[:, start:stop]
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            ctx.link["g"] = g                # frames start:stop are stale until _TakeFramesFn.backward overwrites them
            return g, gm, None, None, None
        if ctx.needs_input_grad[0]:
            g = g.clone()
            g[:, start:stop].zero_()
            return g, None, None, None, None
        return None, gm, None, None, None


class SwinTransformerLayerv5(nn.Module):
    """6 temporal Swin layers over a 4-frame clip in two stages; swin_512.py:280-327.

    forward: (B, 4, C, H, W) -> ((B, 4, C, H, W), (B, 4, 2C, H/2, W/2)).  Outputs are channels-last views of the
    token tensors (no NCHW transpose is materialised); values equal the reference's.
    """

    def __init__(self, dim=512, input_resolution=(64, 80), num_heads=4):
        super().__init__()
        self.dim = dim
        self.input_resolution = tuple(input_resolution)
        self.num_heads = num_heads
        self.num_layers = 3
        self.pairs = [[slice(0, 2), slice(2, 4)], [slice(1, 3)], [slice(0, 2), slice(2, 4)]]
        H, W = self.input_resolution
        self.layers = nn.ModuleList()
        for _ in range(self.num_layers):
            self.layers.append(nn.Sequential(SwinTransformerBlock(dim, (H, W), num_heads),
                                             SwinTransformerBlock(dim, (H, W), num_heads, shift_size=4)))
        for _ in range(self.num_layers):
            self.layers.append(nn.Sequential(
                SwinTransformerBlock(dim * 2, (H // 2, W // 2), num_heads, window_size=4),
                SwinTransformerBlock(dim * 2, (H // 2, W // 2), num_heads, window_size=4, shift_size=2)))
        self.downsample = PatchMerging(self.input_resolution, dim)

    def _single_layer_forward(self, x_v, pairs, layer_idx, own_input=False):
        """Frames inside `pairs` are replaced by layer(pair); the others pass through (swin_512.py:302-307).
        The two disjoint pairs of layers 0/2 share weights, so they run as ONE call with the pair as batch."""
        layer = self.layers[layer_idx]
        B, T, L, C = x_v.shape
        if len(pairs) == 2:
            return layer(x_v.reshape(B * 2, 2, L, C)).reshape(B, T, L, C)
        p = pairs[0]
        if not (torch.is_grad_enabled() and x_v.requires_grad and _FRAME_GRAD_LINK):
            mid = layer(x_v[:, p].contiguous())
            if own_input and not torch.is_grad_enabled():   # a temporary of forward_tokens, no graph: the pair goes back in place
                x_v[:, p].copy_(mid)                        # (momentum-key passes, evaluation: no cat over the whole clip)
                return x_v
            return torch.cat([x_v[:, :p.start], mid.to(x_v.dtype), x_v[:, p.stop:]], dim=1)
        link = {}
        x_v = _FenceFn.apply(x_v)            # exactly two consumers below, whatever else uses the caller's tensor
        mid = layer(_TakeFramesFn.apply(x_v, p.start, p.stop, link))
        return _PutFramesFn.apply(x_v, mid.to(x_v.dtype), p.start, p.stop, link)

    def _stage(self, x, first):
        """Layers first, first + 1, first + 2 on clip tokens (B, 4, L, C): pairs (0,1),(2,3) -> middle pair (1,2) -> pairs (0,1),(2,3)
        (swin_512.py:296-307) WITHOUT the reference's clones / cats of the clip: the first layer writes its output into rows
        [0, 4BL) of one buffer, the middle-pair layer gathers its frames from there through a row map and writes its output into rows
        [4BL, 6BL), and the last layer gathers frames 0, 3 from the first block of rows and frames 1, 2 from the second
        (ops.pair_maps / ops.PairSource; the window gather of a Swin block is a row map anyway, so this is a composition of maps)."""
        B, T, L, C = x.shape
        la, lb, lc = self.layers[first], self.layers[first + 1], self.layers[first + 2]
        dt = ops.compute_dtype(x)
        Z = torch.empty(6 * B * L, C, dtype=dt, device=x.device)
        xmap_mid, xmap_out = ops.pair_maps(B, L, x.device)
        link = {} if torch.is_grad_enabled() else None
        xa = la[1](la[0](x.reshape(B * 2, 2, L, C)), out=Z[:4 * B * L])                          # (2B, 2, L, C) = rows [0, 4BL) of Z
        mid = lb[1](lb[0](xa.reshape(4 * B * L, C), src=ops.PairSource(xmap_mid, link, owner=False)), out=Z[4 * B * L:])
        zj = ops.JoinRowsFn.apply(Z, xa, mid)
        y = lc[1](lc[0](zj, src=ops.PairSource(xmap_out, link, owner=True, publish_rows=4 * B * L)))
        return y.reshape(B, T, L, C)

    def forward_tokens(self, x):
        """(B, 4, L, C) tokens -> ((B, 4, L, C), (B, 4, L/4, 2C)) tokens."""
        B, T, L, C = x.shape
        assert T == 4, "input feature has wrong size"
        x = x.to(ops.compute_dtype(x))
        if _ZERO_COPY_PAIRS:
            out1 = self._stage(x, 0)
            return out1, self._stage(self.downsample(out1), 3)
        for i in range(3):
            x = self._single_layer_forward(x, self.pairs[i], i, own_input=i > 0)       # (i > 0: x is the previous layer's output)
        out1 = x
        x = self.downsample(x)
        for i in range(3):
            x = self._single_layer_forward(x, self.pairs[i], 3 + i, own_input=True)
        return out1, x

    def forward(self, x_v):
        B, T, C, H, W = x_v.shape
        assert T == 4, "input feature has wrong size"
        x = x_v.permute(0, 1, 3, 4, 2).contiguous().view(B, T, H * W, C)
        o1, o2 = self.forward_tokens(x)
        return (o1.view(B, T, H, W, C).permute(0, 1, 4, 2, 3),
                o2.view(B, T, H // 2, W // 2, 2 * C).permute(0, 1, 4, 2, 3))
