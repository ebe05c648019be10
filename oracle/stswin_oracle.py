"""CPU oracle for the STswinCL hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain fp32 PyTorch-CPU restatement of the reference algorithm for every row of
SURVEY.md section 8(a).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the shipped package
``stswincl_amd`` never does (its ops raise when the HIP library is missing).

Style: pure functions over a flat ``state-dict`` (``sd``) whose keys are exactly the
reference's ``nn.Module.state_dict()`` keys, addressed through a string ``prefix``.
That keeps the oracle structurally independent of the reference's class layout
while staying loadable from the same checkpoints.

Parity status: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), so this oracle is pinned against outputs of the reference
itself, generated in the build container by ``tools/gen_golden.py`` (which imports
``/root/reference`` with shims) and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every one of them.

All ``file:line`` citations are relative to ``/root/reference/``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# --------------------------------------------------------------------------------------
# a1-a4: window partition / reverse / cyclic shift / pair regroup  (pure indexing)
# --------------------------------------------------------------------------------------


def window_partition(x: Tensor, ws: int) -> Tensor:
    """(B', H, W, C) -> (B'*nW, ws, ws, C).  seg18/net/Ours/swin_512.py:26-38."""
    b, h, w, c = x.shape
    x = x.reshape(b, h // ws, ws, w // ws, ws, c)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c)


def window_reverse_t(windows: Tensor, ws: int, h: int, w: int, t: int) -> Tensor:
    """(B*nW, T, ws*ws, C) -> (B, T, H, W, C).  swin_512.py:57-71 (T-aware variant)."""
    n_w = (h // ws) * (w // ws)
    b = windows.shape[0] // n_w
    x = windows.reshape(b, h // ws, w // ws, t, ws, ws, -1)
    return x.permute(0, 3, 1, 4, 2, 5, 6).reshape(b, t, h, w, -1)


def pair_window_gather(x_v: Tensor, h: int, w: int, ws: int, shift: int) -> Tensor:
    """(B, T, H*W, C) -> (B*nW, T, ws*ws, C): roll(-s,-s) + partition + pair regroup.

    swin_512.py:207-218 (cyclic shift :210-213, partition :216, regroup :217-218).
    """
    b, t, l, c = x_v.shape
    x = x_v.reshape(b * t, h, w, c)
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(x, ws).reshape(b, t, -1, ws * ws, c)
    return xw.permute(0, 2, 1, 3, 4).reshape(-1, t, ws * ws, c)


def pair_window_scatter(win: Tensor, h: int, w: int, ws: int, shift: int) -> Tensor:
    """(B*nW, T, ws*ws, C) -> (B, T, H*W, C): window_reverse + roll(+s,+s).  swin_512.py:224-231."""
    t = win.shape[1]
    x = window_reverse_t(win, ws, h, w, t)  # B T H W C
    b = x.shape[0]
    x = x.reshape(b * t, h, w, -1)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    return x.reshape(b, t, h * w, -1)


def pair_window_index(b: int, t: int, h: int, w: int, ws: int, shift: int) -> np.ndarray:
    """Flat source-token index (into a (B*T*H*W) token list) for every gathered row.

    Row order is that of ``pair_window_gather``'s output flattened to (B*nW*T*ws*ws).
    Built by pushing an arange through the oracle ops, so it is the *definition* of
    the bit-exact gather the HIP kernels must reproduce.
    """
    ids = torch.arange(b * t * h * w, dtype=torch.int64).reshape(b, t, h * w, 1)
    return pair_window_gather(ids, h, w, ws, shift).reshape(-1).numpy().astype(np.int32)


def patch_merge_gather(x: Tensor, h: int, w: int) -> Tensor:
    """(B, T, H*W, C) -> (B*T, H/2*W/2, 4C) in the order [(0,0),(1,0),(0,1),(1,1)].  swin_512.py:265-272."""
    b, t, l, c = x.shape
    x = x.reshape(b * t, h, w, c)
    parts = [x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]]
    return torch.cat(parts, dim=-1).reshape(b * t, -1, 4 * c)


# --------------------------------------------------------------------------------------
# a5 / a7: constructor-time integer tables
# --------------------------------------------------------------------------------------


def relative_position_index(ws: int) -> Tensor:
    """(ws*ws, ws*ws) int64 index into the (2ws-1)^2 bias table.  swin_512.py:89-99."""
    ch, cw = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    flat = torch.stack([ch.reshape(-1), cw.reshape(-1)])  # 2, N
    rel = flat[:, :, None] - flat[:, None, :]  # 2, N, N
    return (rel[0] + ws - 1) * (2 * ws - 1) + (rel[1] + ws - 1)


def shift_attn_mask(h: int, w: int, ws: int, shift: int) -> Optional[Tensor]:
    """(nW, ws*ws, ws*ws) in {0,-100} for SW-MSA; None when shift == 0.  swin_512.py:171-194."""
    if shift <= 0:
        return None
    img = torch.zeros(1, h, w, 1)
    bounds_h = [(0, h - ws), (h - ws, h - shift), (h - shift, h)]
    bounds_w = [(0, w - ws), (w - ws, w - shift), (w - shift, w)]
    region = 0
    for h0, h1 in bounds_h:
        for w0, w1 in bounds_w:
            img[:, h0:h1, w0:w1, :] = region
            region += 1
    mw = window_partition(img, ws).reshape(-1, ws * ws)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def effective_window(res: Tuple[int, int], ws: int, shift: int) -> Tuple[int, int]:
    """swin_512.py:155-158: no partitioning when the map is not larger than the window."""
    if min(res) <= ws:
        return min(res), 0
    return ws, shift


# --------------------------------------------------------------------------------------
# a6: window attention
# --------------------------------------------------------------------------------------


def expanded_rel_bias(sd: SD, prefix: str, ws: int, heads: int) -> Tensor:
    """(heads, N, N) bias = table[index].  swin_512.py:122-124 (before the (T,T) tiling)."""
    table = sd[prefix + "relative_position_bias_table"]
    index = sd.get(prefix + "relative_position_index")
    if index is None:
        index = relative_position_index(ws)
    n = ws * ws
    return table[index.reshape(-1).long()].reshape(n, n, heads).permute(2, 0, 1).contiguous()


def window_attention(x_v: Tensor, sd: SD, prefix: str, heads: int, ws: int, mask: Optional[Tensor]) -> Tensor:
    """x_v (B_, T, N, C) -> (B_, T, N, C).  swin_512.py:109-141.

    q*scale ; q k^T ; + bias tiled (T,T) ; + mask tiled (T,T) per window ; softmax ; @ v ; proj.
    """
    b_, t, n, c = x_v.shape
    d = c // heads
    qkv = F.linear(x_v.reshape(-1, c), sd[prefix + "qkv.weight"], sd.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(b_, t * n, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (d ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)  # B_, h, TN, TN
    bias = expanded_rel_bias(sd, prefix, ws, heads).repeat(1, t, t)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        n_w = mask.shape[0]
        m = mask.repeat(1, t, t)
        attn = attn.reshape(b_ // n_w, n_w, heads, t * n, t * n) + m[None, :, None]
        attn = attn.reshape(-1, heads, t * n, t * n)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(b_, t, n, c)
    return F.linear(out, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


# --------------------------------------------------------------------------------------
# a8 / a9: block (post-norm topology) and MLP
# --------------------------------------------------------------------------------------


def mlp(x: Tensor, sd: SD, prefix: str) -> Tensor:
    """fc1 -> GELU(erf) -> fc2.  swin_512.py:7-23."""
    x = F.linear(x, sd[prefix + "fc1.weight"], sd[prefix + "fc1.bias"])
    x = F.gelu(x)
    return F.linear(x, sd[prefix + "fc2.weight"], sd[prefix + "fc2.bias"])


def layer_norm(x: Tensor, sd: SD, prefix: str, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"], sd[prefix + "bias"], eps)


def swin_block(x_v: Tensor, sd: SD, prefix: str, res: Tuple[int, int], heads: int, ws: int, shift: int) -> Tensor:
    """(B, 2, L, C) -> (B, 2, L, C).  swin_512.py:196-237.

    NB the reference's odd topology: no pre-attention norm (:205 commented out),
    ``x = shortcut + attn`` (:234) then ``x = norm1(x + mlp(norm2(x)))`` (:235).
    """
    h, w = res
    ws, shift = effective_window(res, ws, shift)
    b, t, l, c = x_v.shape
    assert l == h * w and t == 2
    mask = sd.get(prefix + "attn_mask") if shift > 0 else None
    if shift > 0 and mask is None:
        mask = shift_attn_mask(h, w, ws, shift)
    win = pair_window_gather(x_v, h, w, ws, shift)
    att = window_attention(win, sd, prefix + "attn.", heads, ws, mask)
    x = x_v + pair_window_scatter(att, h, w, ws, shift)
    y = x + mlp(layer_norm(x, sd, prefix + "norm2."), sd, prefix + "mlp.")
    return layer_norm(y, sd, prefix + "norm1.")


# --------------------------------------------------------------------------------------
# a10 / a11: patch merging and the 6-layer temporal schedule
# --------------------------------------------------------------------------------------


def patch_merging(x: Tensor, sd: SD, prefix: str, res: Tuple[int, int]) -> Tensor:
    """(B, 4, L, C) -> (B, 4, L/4, 2C).  swin_512.py:255-277."""
    b, t, l, c = x.shape
    assert t == 4 and l == res[0] * res[1]
    g = patch_merge_gather(x, res[0], res[1])
    g = layer_norm(g, sd, prefix + "norm.")
    g = F.linear(g, sd[prefix + "reduction.weight"])
    return g.reshape(b, t, l // 4, 2 * c)


PAIR_SCHEDULE: List[List[Tuple[int, int]]] = [[(0, 2), (2, 4)], [(1, 3)], [(0, 2), (2, 4)]]  # swin_512.py:287


def _temporal_layer(x: Tensor, sd: SD, prefix: str, pairs, res, heads, ws, shift) -> Tensor:
    """swin_512.py:302-307: clone, then overwrite each frame pair with layer(pair)."""
    y = x.clone()
    for lo, hi in pairs:
        z = x[:, lo:hi]
        z = swin_block(z, sd, prefix + "0.", res, heads, ws, 0)
        z = swin_block(z, sd, prefix + "1.", res, heads, ws, shift)
        y[:, lo:hi] = z
    return y


def swin_layer_v5(x_v: Tensor, sd: SD, prefix: str, heads: int = 4) -> Tuple[Tensor, Tensor]:
    """(B, 4, C, H, W) -> ((B,4,C,H,W), (B,4,2C,H/2,W/2)).  swin_512.py:309-327.

    Stage 1: window 8 / shift 4 at (H, W); stage 2: window 4 / shift 2 at (H/2, W/2) (:291-298).
    """
    b, t, c, h, w = x_v.shape
    assert t == 4
    x = x_v.permute(0, 1, 3, 4, 2).reshape(b, t, h * w, c)
    for i in range(3):
        x = _temporal_layer(x, sd, f"{prefix}layers.{i}.", PAIR_SCHEDULE[i], (h, w), heads, 8, 4)
    out1 = x.permute(0, 1, 3, 2).reshape(b, t, c, h, w)
    x = patch_merging(x, sd, prefix + "downsample.", (h, w))
    for i in range(3):
        x = _temporal_layer(x, sd, f"{prefix}layers.{3 + i}.", PAIR_SCHEDULE[i], (h // 2, w // 2), heads, 4, 2)
    out2 = x.permute(0, 1, 3, 2).reshape(b, t, 2 * c, h // 2, w // 2)
    return out1, out2


# --------------------------------------------------------------------------------------
# a12: ResNet18-OS8 feeder
# --------------------------------------------------------------------------------------


def batch_norm(x: Tensor, sd: SD, prefix: str, training: bool, momentum: float = 0.1, eps: float = 1e-5) -> Tensor:
    """nn.BatchNorm2d semantics incl. in-place running-stat update in training mode."""
    rm, rv = sd.get(prefix + "running_mean"), sd.get(prefix + "running_var")
    y = F.batch_norm(x, rm, rv, sd[prefix + "weight"], sd[prefix + "bias"], training, momentum, eps)
    if training and (prefix + "num_batches_tracked") in sd:
        sd[prefix + "num_batches_tracked"] += 1
    return y


def _basic_block(x: Tensor, sd: SD, prefix: str, stride: int, dilation: int, training: bool, tv_style: bool) -> Tensor:
    """resnet.py:22-53 (own blocks; ``downsample`` = Sequential(conv, bn) keys .0/.1), and
    the torchvision BasicBlock of the stem layers (same math, same key names)."""
    pad = dilation
    out = F.conv2d(x, sd[prefix + "conv1.weight"], None, stride, pad, dilation)
    out = F.relu(batch_norm(out, sd, prefix + "bn1.", training))
    out = F.conv2d(out, sd[prefix + "conv2.weight"], None, 1, pad, dilation)
    out = batch_norm(out, sd, prefix + "bn2.", training)
    if (prefix + "downsample.0.weight") in sd:
        idn = F.conv2d(x, sd[prefix + "downsample.0.weight"], None, stride)
        idn = batch_norm(idn, sd, prefix + "downsample.1.", training)
    else:
        idn = x
    return F.relu(out + idn)


def resnet18_os8(x: Tensor, sd: SD, prefix: str, training: bool) -> Tensor:
    """(B,3,H,W) -> (B,512,H/8,W/8).  resnet.py:93-133.

    ``resnet.{0,1}`` = conv7x7/2 + BN, relu, maxpool3/2, ``resnet.4`` = layer1 (2xBB 64),
    ``resnet.5`` = layer2 (2xBB 128, /2)  [torchvision children[:-4], resnet.py:102];
    ``layer4`` 2xBB 256 dilation 2, ``layer5`` 2xBB 512 dilation 4 (:117-119).
    """
    p = prefix + "resnet."
    x = F.conv2d(x, sd[p + "0.weight"], None, 2, 3)
    x = F.relu(batch_norm(x, sd, p + "1.", training))
    x = F.max_pool2d(x, 3, 2, 1)
    for blk in range(2):
        x = _basic_block(x, sd, f"{p}4.{blk}.", 1, 1, training, True)
    for blk in range(2):
        x = _basic_block(x, sd, f"{p}5.{blk}.", 2 if blk == 0 else 1, 1, training, True)
    for blk in range(2):
        x = _basic_block(x, sd, f"{prefix}layer4.{blk}.", 1, 2, training, False)
    for blk in range(2):
        x = _basic_block(x, sd, f"{prefix}layer5.{blk}.", 1, 4, training, False)
    return x


# --------------------------------------------------------------------------------------
# a13: ASPP ; a14: TswinPlus assembly
# --------------------------------------------------------------------------------------


def aspp(x: Tensor, sd: SD, prefix: str, training: bool) -> Tensor:
    """(B,1024,h,w) -> (B,256,h,w).  seg18/net/Ours/ASPP.py:33-52 (== ASPPv5)."""
    h, w = x.shape[2:]

    def cbr(inp, conv, bn, dil=None):
        if dil is None:
            y = F.conv2d(inp, sd[prefix + conv + ".weight"], sd[prefix + conv + ".bias"])
        else:
            y = F.conv2d(inp, sd[prefix + conv + ".weight"], sd[prefix + conv + ".bias"], 1, dil, dil)
        return F.relu(batch_norm(y, sd, prefix + bn + ".", training))

    o1 = cbr(x, "conv_1x1_1", "bn_conv_1x1_1")
    o2 = cbr(x, "conv_3x3_1", "bn_conv_3x3_1", 6)
    o3 = cbr(x, "conv_3x3_2", "bn_conv_3x3_2", 12)
    o4 = cbr(x, "conv_3x3_3", "bn_conv_3x3_3", 18)
    img = F.adaptive_avg_pool2d(x, 1)
    img = cbr(img, "conv_1x1_2", "bn_conv_1x1_2")
    img = F.interpolate(img, size=(h, w), mode="bilinear", align_corners=False)
    cat = torch.cat([o1, o2, o3, o4, img], 1)
    out = cbr(cat, "conv_1x1_3", "bn_conv_1x1_3")
    return F.conv2d(out, sd[prefix + "conv_1x1_4.weight"], sd[prefix + "conv_1x1_4.bias"])


def _project(x: Tensor, sd: SD, prefix: str, training: bool) -> Tensor:
    """1x1 conv (no bias) + BN + ReLU.  base18.py:60-71."""
    y = F.conv2d(x, sd[prefix + "0.weight"])
    return F.relu(batch_norm(y, sd, prefix + "1.", training))


def decode_features(x: Tensor, sd: SD, prefix: str, training: bool) -> Tensor:
    """Frames (B,4,3,H,W) -> the 400-channel concat at (H/8, W/8).  base18.py:80-105 minus classifier.

    The ResNet runs once per frame (:86-89), so train-mode BN statistics are per frame.
    """
    b, t = x.shape[:2]
    feats = [resnet18_os8(x[:, i], sd, prefix + "resnet.", training) for i in range(t)]
    tem = torch.stack(feats, dim=1)
    res_last = tem[:, -1]
    tem1, tem2 = swin_layer_v5(tem, sd, prefix + "swin.")
    t1, t2 = tem1[:, -1], tem2[:, -1]
    a = aspp(t2, sd, prefix + "aspp.", training)
    p1 = _project(res_last, sd, prefix + "project1.", training)
    p2 = _project(t1, sd, prefix + "project2.", training)
    p3 = _project(t2, sd, prefix + "project3.", training)
    size = p1.shape[2:]
    p3 = F.interpolate(p3, size=size, mode="bilinear", align_corners=False)
    a = F.interpolate(a, size=size, mode="bilinear", align_corners=False)
    return torch.cat([p1, p2, p3, a], dim=1)


def tswin_plus(x: Tensor, sd: SD, training: bool, prefix: str = "") -> Tensor:
    """(B,4,3,H,W) -> logits (B,nc,H,W).  base18.py:80-108."""
    hh, ww = x.shape[3:]
    f = decode_features(x, sd, prefix, training)
    y = F.conv2d(f, sd[prefix + "classifier.0.weight"], None, 1, 1)
    y = F.relu(batch_norm(y, sd, prefix + "classifier.1.", training))
    y = F.conv2d(y, sd[prefix + "classifier.3.weight"], sd[prefix + "classifier.3.bias"])
    return F.interpolate(y, (hh, ww), mode="bilinear", align_corners=False)


# --------------------------------------------------------------------------------------
# a15: OHEM cross entropy
# --------------------------------------------------------------------------------------


def ohem_ce(logits: Tensor, labels: Tensor, n_min: int, thresh: float = 0.7, ignore_index: int = -1) -> Tensor:
    """seg18/utils/losses.py:16-40.  Per-pixel CE, sort desc, keep >thresh or top n_min, mean."""
    t = -math.log(thresh)
    loss = F.cross_entropy(logits, labels, ignore_index=ignore_index, reduction="none").reshape(-1)
    loss, _ = torch.sort(loss, descending=True)
    if loss[n_min] > t:
        kept = loss[loss > t]
    else:
        kept = loss[:n_min]
    return kept.mean()


# --------------------------------------------------------------------------------------
# a16 / a17: label-guided pixel-contrastive loss
# --------------------------------------------------------------------------------------


def pos_mask(lab_a: Tensor, lab_b: Tensor) -> Tensor:
    """(B,1,h,w) x2 -> (B,HW,HW) label-equality matrix.  PixPro_swin_v5.py:48-57.

    The reference builds it as bmm(one_hot, one_hot^T); for integer labels in [0,nc) that is
    exactly ``label_i == label_j``.
    """
    b = lab_a.shape[0]
    a = lab_a.reshape(b, -1).long()
    c = lab_b.reshape(b, -1).long()
    return (a[:, :, None] == c[:, None, :]).float()


def regression_loss(q, k, adj1, adj2, adj3, neg3, l_q, l_k, l_a1, l_a2, l_a3, l_n3, class_num: int = 12) -> Tensor:
    """PixPro_swin_v5.py:71-129.

    P_i = sum_j sum_pos logit / (sum_j |pos| + 1e-6);  N_i = sum_j [sum_neg logit_j / (|neg_j| + 1e-6)];
    loss = -mean(log(e^P / (e^P + e^N) + 1e-6)).  No temperature, no softmax over keys.
    """
    n, c = q.shape[:2]
    qf = q.reshape(n, c, -1).transpose(1, 2)  # N, HW, C
    keys = [k, adj1, adj2, adj3, neg3]
    labs = [l_k, l_a1, l_a2, l_a3, l_n3]
    pos_sum = 0.0
    pos_cnt = 0.0
    neg_term = 0.0
    for key, lab in zip(keys, labs):
        logit = torch.bmm(qf, key.reshape(n, c, -1))
        mp = pos_mask(l_q, lab)
        mn = 1.0 - mp
        pos_sum = pos_sum + (mp * logit).sum(-1)
        pos_cnt = pos_cnt + mp.sum(-1)
        neg_term = neg_term + (mn * logit).sum(-1) / (mn.sum(-1) + 1e-6)
    p = pos_sum / (pos_cnt + 1e-6)
    pe, ne = torch.exp(p), torch.exp(neg_term)
    return -torch.mean(torch.log(pe / (pe + ne) + 1e-6))


def mlp2d(x: Tensor, sd: SD, prefix: str, training: bool) -> Tensor:
    """1x1 conv + BN + ReLU + 1x1 conv.  PixPro_swin_v5.py:29-46."""
    y = F.conv2d(x, sd[prefix + "linear1.weight"], sd[prefix + "linear1.bias"])
    y = F.relu(batch_norm(y, sd, prefix + "bn1.", training))
    return F.conv2d(y, sd[prefix + "linear2.weight"], sd[prefix + "linear2.bias"])


# --------------------------------------------------------------------------------------
# a18-a20: PixPro encoders, EMA, ConsistencyLoss
# --------------------------------------------------------------------------------------

_Q_PARTS = [("encoder_1.", "resnet."), ("encoder_2.", "swin."), ("encoder_3.", "aspp."),
            ("proj1.", "project1."), ("proj2.", "project2."), ("proj3.", "project3.")]
_K_PARTS = [("encoder_k_1.", "resnet."), ("encoder_k_2.", "swin."), ("encoder_k_3.", "aspp."),
            ("proj_k_1.", "project1."), ("proj_k_2.", "project2."), ("proj_k_3.", "project3.")]


class _Remap(dict):
    """View of a PixPro state-dict under TswinPlus key names (keeps tensors shared, so BN
    running-stat updates land in the underlying dict)."""

    def __init__(self, sd: SD, root: str, parts):
        super().__init__()
        for src, dst in parts:
            pre = root + src
            for key, val in sd.items():
                if key.startswith(pre):
                    self[dst + key[len(pre):]] = val


def pixel_embed(x: Tensor, sd: SD, root: str, key_branch: bool, training: bool = True) -> Tensor:
    """One encoder pass: frames (B,4,3,h,w) -> L2-normalised (B,256,h/8,w/8).  PixPro_swin_v5.py:302-331."""
    view = _Remap(sd, root, _K_PARTS if key_branch else _Q_PARTS)
    f = decode_features(x, view, "", training)
    proj = mlp2d(f, sd, root + ("projector_k." if key_branch else "projector."), training)
    return F.normalize(proj, dim=1)


def ema_momentum(k: int, big_k: int, base: float = 0.99) -> float:
    """PixPro_swin_v5.py:262: 1 - (1-m)(cos(pi k/K)+1)/2."""
    return 1.0 - (1.0 - base) * (math.cos(math.pi * k / big_k) + 1.0) / 2.0


_EMA_PAIRS = [("encoder_1.", "encoder_k_1."), ("encoder_2.", "encoder_k_2."), ("encoder_3.", "encoder_k_3."),
              ("proj1.", "proj_k_1."), ("proj2.", "proj_k_2."), ("proj3.", "proj_k_3."),
              ("projector.", "projector_k.")]


def momentum_update(sd: SD, root: str, param_keys: Sequence[str], m: float) -> None:
    """k <- k*m + q*(1-m) over *parameters* (not buffers).  PixPro_swin_v5.py:258-289.

    ``param_keys``: the query-side parameter names (relative to ``root``), i.e. what
    ``module.parameters()`` would enumerate; buffers (BN running stats, attn_mask,
    relative_position_index) are not averaged by the reference.
    """
    for qk in param_keys:
        for qpre, kpre in _EMA_PAIRS:
            if qk.startswith(qpre):
                kk = root + kpre + qk[len(qpre):]
                sd[kk] = sd[kk] * m + sd[root + qk] * (1.0 - m)
                break


def consistency_loss(ims: Sequence[Tensor], masks: Sequence[Tensor], sd: SD, param_keys: Sequence[str],
                     k: int, big_k: int, class_num: int = 12, root: str = "pixpro.") -> Tuple[Tensor, int]:
    """PixPro_swin_v5.py:291-561 + :578-597.  Returns (loss, k+1).

    2 query passes (seq_1, seq_2), EMA, 6 key passes (seq_1..seq_6, all BN in train mode),
    nearest-resize of the 6 label maps, then two symmetric ``regression_loss`` terms.
    """
    pred1 = pixel_embed(ims[0], sd, root, False)
    pred2 = pixel_embed(ims[1], sd, root, False)
    with torch.no_grad():
        momentum_update(sd, root, param_keys, ema_momentum(k, big_k))
        keys = [pixel_embed(im, sd, root, True) for im in ims]
    hh, ww = pred1.shape[2:]
    lm = [F.interpolate(m, size=[hh, ww], mode="nearest") for m in masks]
    loss = regression_loss(pred1, keys[1], keys[2], keys[3], keys[4], keys[5],
                           lm[0], lm[1], lm[2], lm[3], lm[4], lm[5], class_num) \
        + regression_loss(pred2, keys[0], keys[2], keys[3], keys[4], keys[5],
                          lm[1], lm[0], lm[2], lm[3], lm[4], lm[5], class_num)
    return loss, k + 1


# --------------------------------------------------------------------------------------
# f2: LARS over SGD-momentum (the contrastive stage's optimizer)
# --------------------------------------------------------------------------------------


def lars_sgd_step(params: Sequence[Tensor], grads: Sequence[Tensor], bufs: List[Optional[Tensor]], lr: float, momentum: float,
                  weight_decay: float, adaptive: bool, trust_coef: float = 0.001, eps: float = 1e-8) -> None:
    """One step of pixcontrast_18/contrast/lars.py:109-152 around torch.optim.SGD (momentum, dampening 0) for ONE parameter
    group, in place on `params` / `bufs`: weight decay joins the gradient first (:124-125), the trust ratio
    trust_coef * |p| / (|g| + eps) scales it only in groups with ignore == False and only when both norms are > 0 (:129-142),
    then the inner SGD runs with weight decay 0 (:147-150): buf = g on a parameter's first step, momentum * buf + g after."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if weight_decay > 0:
            g = g + weight_decay * p
        if adaptive:
            pn, gn = p.norm(), g.norm()
            if pn > 0 and gn > 0:
                g = g * (trust_coef * pn / (gn + eps))
        bufs[i] = g.clone() if bufs[i] is None else bufs[i] * momentum + g
        p.sub_(lr * bufs[i])


# --------------------------------------------------------------------------------------
# N1: bank mode of the pixel-contrastive loss (dense restatement)
# --------------------------------------------------------------------------------------


def bank_scores(q: Tensor, lq: Tensor, bank: Tensor, lb: Tensor, gmap: Sequence[Sequence[int]], q_block: int, bank_block: int,
                inv_tau: float = 1.0) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """q [M][C] with labels lq [M]: `len(gmap)` equal query sets of blocks of q_block rows; bank [maps][seg][C] with labels lb;
    query set s uses map gmap[s][g] as group g, block b sees bank rows [b*bank_block, (b+1)*bank_block) (every row if the
    bank has a single block).  Materialises the logits like PixPro_swin_v5.py:82-113 does per sample (logit = bmm(q^T, key),
    posMask = label equality, negMask = 1 - posMask) and returns pos, neg [M][G] (the two masked row sums), rowmax, lse [M] (of
    inv_tau * logit over all groups).  With gmap = [[0..4]] and blocks = samples this is exactly the five bmm + mask
    products of the reference (an empty negative set contributes an exactly-zero sum AND gradient, as there)."""
    M, _ = q.shape
    groups = len(gmap[0])
    per_set = M // len(gmap)
    nb = bank.shape[1] // bank_block
    pos_rows: List[Tensor] = []
    neg_rows: List[Tensor] = []
    rows_logits: List[Tensor] = []
    for s, maps in enumerate(gmap):
        for b in range(per_set // q_block):
            r0 = s * per_set + b * q_block
            sl = slice(r0, r0 + q_block)
            bb = b if nb > 1 else 0
            chunks, pg, ng = [], [], []
            for g, mp in enumerate(maps):
                keys = bank[mp, bb * bank_block:(bb + 1) * bank_block]
                logit = q[sl] @ keys.t()
                mask = (lq[sl, None] == lb[mp, None, bb * bank_block:(bb + 1) * bank_block]).to(logit.dtype)
                pg.append((logit * mask).sum(1))
                ng.append((logit * (1.0 - mask)).sum(1))
                chunks.append(logit)
            pos_rows.append(torch.stack(pg, 1))
            neg_rows.append(torch.stack(ng, 1))
            rows_logits.append(torch.cat(chunks, 1) * inv_tau)
    allv = torch.cat(rows_logits, 0)
    return torch.cat(pos_rows, 0), torch.cat(neg_rows, 0), allv.max(1).values, torch.logsumexp(allv, 1)


def bank_contrast_loss(q: Tensor, lq: Tensor, bank: Tensor, lb: Tensor, gmap, q_block: int, bank_block: int) -> Tensor:
    """PixPro_swin_v5.py:114-129 on bank_scores: P = sum_g pos / (sum_g |pos| + 1e-6), N = sum_g neg_g / (|neg_g| + 1e-6),
    loss = sum over query sets of -mean(log(e^P / (e^P + e^N) + 1e-6))."""
    pos, neg, _, _ = bank_scores(q, lq, bank, lb, gmap, q_block, bank_block)
    M = q.shape[0]
    per_set = M // len(gmap)
    nb = bank.shape[1] // bank_block
    cnt = torch.zeros_like(pos)
    for m in range(M):
        s, b = m // per_set, ((m % per_set) // q_block if nb > 1 else 0)
        for g, mp in enumerate(gmap[s]):
            cnt[m, g] = float((lb[mp, b * bank_block:(b + 1) * bank_block] == lq[m]).sum())
    P = pos.sum(1) / (cnt.sum(1) + 1e-6)
    N = (neg / ((bank_block - cnt) + 1e-6)).sum(1)
    term = -torch.log(torch.exp(P) / (torch.exp(P) + torch.exp(N)) + 1e-6)
    return term.view(len(gmap), -1).mean(1).sum()
