"""MI355X-native drop-in for ``contrast.models.PixPro_swin_v5`` (pixcontrast_18/contrast/models/PixPro_swin_v5.py).

Same public names and signatures: ``MLP2d``, ``posMask``, ``negMask``, ``regression_loss`` (13 positional args),
``Proj_Head``, ``Pred_Head``, ``PixPro`` (children encoder_1/2/3, proj1/2/3, projector, encoder_k_*, proj_k_*,
projector_k, value_transform), ``ConsistencyLoss(args).forward(im_1..im_6, mask_1..mask_6)``.

The similarity / label-mask reductions run in one HIP kernel (csrc/contrast.hip); the encoders are the
TswinPlus pipeline of stswincl_amd.net.Ours on NHWC tokens.  No ``.cuda()`` calls are hard-coded: modules follow
the device they are moved to (which must be the GPU).
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import headops as H
from ... import hip
from ...ops import compute_dtype
from .Ours.base import TswinPlusv5, decode_tokens

num_class_table = {'1': 9, '2': 18, '3': 26}   # CaDIS tags incl. the ignore class (PixPro_swin_v5.py:14)


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class Identity(nn.Module):
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return input


def conv1x1(in_planes, out_planes):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=1, padding=0, bias=True)


class MLP2d(nn.Module):
    """1x1 conv + BN + ReLU + 1x1 conv (PixPro_swin_v5.py:29-46)."""

    def __init__(self, in_dim, inner_dim=4096, out_dim=256):
        super().__init__()
        self.linear1 = conv1x1(in_dim, inner_dim)
        self.bn1 = nn.BatchNorm2d(inner_dim)
        self.relu1 = nn.ReLU(inplace=True)
        self.linear2 = conv1x1(inner_dim, out_dim)

    def forward_tokens(self, x, geom, lin=None):
        f, h, w = geom
        y = H.conv_bn_relu(x, self.linear1, self.bn1, geom, lin=lin)
        return H.conv1x1_tokens(y, self.linear2, f, h, w)

    def forward(self, x):
        f, c, h, w = x.shape
        lin = H.Layout.dense(c)
        out = self.forward_tokens(H.pad_cols(H.to_tokens(x), lin.width), (f, h, w), lin)
        return H.from_tokens(out, f, h, w)[:, :self.linear2.out_channels]


def posMask(pred1, pred2, class_num):
    """(B,1,H,W) label maps -> (B,HW,HW) equality matrix (PixPro_swin_v5.py:48-57).  Provided for API parity only:
    regression_loss never materialises it."""
    B = pred1.shape[0]
    a, b = pred1.reshape(B, -1).long(), pred2.reshape(B, -1).long()
    return (a[:, :, None] == b[:, None, :]).float()


def negMask(pred1, pred2, class_num):
    return 1 - posMask(pred1, pred2, class_num)


class ContrastSumsFn(torch.autograd.Function):
    """(q, 5 key maps, labels) -> pos[N,HW,5], neg[N,HW,5]: masked similarity sums over label-equal / label-different
    key pixels (see csrc/contrast.hip).  Keys are no-grad in the reference (PixPro_swin_v5.py:366).  Backward to q uses
    sum_p [l_i == l_p] k_p = Kpos[l_i] and sum_p [l_i != l_p] k_p = Kneg[l_i] (per-class sums of the keys, built with
    exact 0/1 masks so an empty set gives an exactly-zero gradient like the reference's masked products):
    dq_i = sum_j dpos_ij Kpos_j[l_i] + dneg_ij Kneg_j[l_i] - the gradient of the dense products at O(HW C) cost."""

    @staticmethod
    def forward(ctx, q, k0, k1, k2, k3, k4, lq, lk, N, HW, class_num):
        dt = compute_dtype(q)
        Q = q.detach().to(dt).contiguous()
        keys = [k.detach().to(dt).contiguous() for k in (k0, k1, k2, k3, k4)]
        pos, tot = hip.contrast_fwd(Q, keys, lq, [lk[j] for j in range(5)], N, HW)
        ctx.cfg = (N, HW, class_num, q.dtype)
        ctx.save_for_backward(lq, lk, *keys)
        return pos, tot - pos

    @staticmethod
    def backward(ctx, dpos, dneg):
        lq, lk, *keys = ctx.saved_tensors
        N, HW, ncls, in_dtype = ctx.cfg
        C = keys[0].shape[1]
        dq = torch.zeros(N, HW, C, dtype=torch.float32, device=keys[0].device)
        idx = lq.long().unsqueeze(-1).expand(N, HW, C)
        for j, k in enumerate(keys):
            kf = k.float().view(N, HW, C)
            oh = F.one_hot(lk[j].long(), ncls).float()                      # [N][HW][cls], exact 0/1
            kpos = torch.bmm(oh.transpose(1, 2), kf)                        # [N][cls][C]
            kneg = torch.bmm((1.0 - oh).transpose(1, 2), kf)
            dq += dpos[:, :, j:j + 1] * torch.gather(kpos, 1, idx) + dneg[:, :, j:j + 1] * torch.gather(kneg, 1, idx)
        return (dq.view(N * HW, C).to(in_dtype),) + (None,) * 10


def regression_loss(q, k, adj1, adj2, adj3, neg3, label_patch1, label_patch2, label_adj1, label_adj2, label_adj3,
                    label_neg3, class_num):
    """PixPro_swin_v5.py:71-129.  P = sum_j sum_pos / (sum_j |pos| + 1e-6); N = sum_j sum_neg_j / (|neg_j| + 1e-6);
    loss = -mean(log(e^P / (e^P + e^N) + 1e-6)).  No temperature, no softmax over keys."""
    n, c, h, w = q.shape
    HW = h * w
    labs = [l.reshape(n, HW).to(torch.int32).contiguous() for l in
            (label_patch1, label_patch2, label_adj1, label_adj2, label_adj3, label_neg3)]
    lq, lk = labs[0], torch.stack(labs[1:], 0)
    toks = [H.to_tokens(t) for t in (q, k, adj1, adj2, adj3, neg3)]
    pos, neg = ContrastSumsFn.apply(toks[0], *toks[1:], lq, lk, n, HW, class_num)
    hist = F.one_hot(lk.long(), class_num).sum(2).float()                       # [5][N][cls]
    cnt = torch.gather(hist, 2, lq.long().unsqueeze(0).expand(5, n, HW)).permute(1, 2, 0)   # [N][HW][5]
    P = pos.sum(-1) / (cnt.sum(-1) + 1e-6)
    Nn = (neg / ((HW - cnt) + 1e-6)).sum(-1)
    pe, ne = torch.exp(P), torch.exp(Nn)
    return -torch.mean(torch.log(pe / (pe + ne) + 1e-6))


def Proj_Head(in_dim=400, inner_dim=512, out_dim=256):
    return MLP2d(in_dim, inner_dim, out_dim)


def Pred_Head(in_dim=256, inner_dim=4096, out_dim=256):
    return MLP2d(in_dim, inner_dim, out_dim)


LCAT = H.Layout.concat([H.Layout.dense(48)] * 3 + [H.Layout.dense(256)])


def _batched_views_ok(model) -> bool:
    """View batching needs the interleaved-group BatchNorm kernels: local statistics only (SyncBatchNorm across ranks keeps
    the sequential passes), train mode, and not switched off (STSWIN_SEQUENTIAL_VIEWS=1: A/B runs)."""
    if os.environ.get("STSWIN_SEQUENTIAL_VIEWS") == "1" or not model.training:
        return False
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return not any(isinstance(m, nn.SyncBatchNorm) for m in model.modules())
    return True


class PixPro(nn.Module):
    """Query / momentum-key encoders (PixPro_swin_v5.py:140-561)."""

    def __init__(self, args, input_resolution=(32, 56)):
        super().__init__()
        self.pixpro_p = args.pixpro_p
        self.pixpro_momentum = args.pixpro_momentum
        self.pixpro_clamp_value = args.pixpro_clamp_value
        self.pixpro_transform_layer = args.pixpro_transform_layer
        self.pixpro_ins_loss_weight = args.pixpro_ins_loss_weight
        if args.data == 'endo18':
            class_num = 12
        elif args.data == 'cata':
            class_num = int(num_class_table[args.tag])
        else:
            raise NotImplementedError(args.data)
        if self.pixpro_ins_loss_weight > 0.:
            raise NotImplementedError("instance branch is disabled in the reference's scripts (pixpro_ins_loss_weight 0)")
        seg_q, seg_k = TswinPlusv5(class_num, input_resolution), TswinPlusv5(class_num, input_resolution)
        pre = getattr(args, "pretrainpth", None)
        if pre and os.path.exists(pre):          # the reference reads 'xx/results/' + pretrainpth (:155-166)
            sd = torch.load(pre, map_location="cpu")
            sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
            seg_q.load_state_dict(sd, strict=False)
            seg_k.load_state_dict(sd, strict=False)
        self.encoder_1, self.encoder_2, self.encoder_3 = seg_q.resnet, seg_q.swin, seg_q.aspp
        self.proj1, self.proj2, self.proj3 = seg_q.project1, seg_q.project2, seg_q.project3
        self.projector = Proj_Head()
        self.encoder_k_1, self.encoder_k_2, self.encoder_k_3 = seg_k.resnet, seg_k.swin, seg_k.aspp
        self.proj_k_1, self.proj_k_2, self.proj_k_3 = seg_k.project1, seg_k.project2, seg_k.project3
        self.projector_k = Proj_Head()
        for q_mod, k_mod in self._pairs():
            for pq, pk in zip(q_mod.parameters(), k_mod.parameters()):
                pk.data.copy_(pq.data)
                pk.requires_grad = False
        # SyncBatchNorm over both encoders, as the reference (PixPro_swin_v5.py:215-228); with one process this is
        # ordinary BatchNorm.  The grouped-BN kernels combine the cross-rank statistics (headops.BNTokFn).
        for mod in (self.encoder_1, self.encoder_2, self.encoder_3, self.proj1, self.proj2, self.proj3, self.encoder_k_1,
                    self.encoder_k_2, self.encoder_k_3, self.proj_k_1, self.proj_k_2, self.proj_k_3, self.projector,
                    self.projector_k):
            nn.SyncBatchNorm.convert_sync_batchnorm(mod)
        ws = _world_size()
        self.K = int(args.num_instances * 1. / ws / args.batch_size * args.epochs)
        self.k = int(args.num_instances * 1. / ws / args.batch_size * (args.start_epoch - 1))
        if self.pixpro_transform_layer == 0:
            self.value_transform = Identity()
        elif self.pixpro_transform_layer == 1:
            self.value_transform = conv1x1(in_planes=256, out_planes=256)
        elif self.pixpro_transform_layer == 2:
            self.value_transform = MLP2d(in_dim=256, inner_dim=256, out_dim=256)
        else:
            raise NotImplementedError

    def _pairs(self):
        return [(self.encoder_1, self.encoder_k_1), (self.encoder_2, self.encoder_k_2), (self.encoder_3, self.encoder_k_3),
                (self.proj1, self.proj_k_1), (self.proj2, self.proj_k_2), (self.proj3, self.proj_k_3),
                (self.projector, self.projector_k)]

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """k <- k m + q (1-m), m = 1 - (1-m0)(cos(pi k/K)+1)/2 (PixPro_swin_v5.py:258-289), on the multi-tensor HIP kernel."""
        m = 1. - (1. - self.pixpro_momentum) * (math.cos(math.pi * self.k / self.K) + 1) / 2.
        self.k = self.k + 1
        qs, ks = [], []
        for q_mod, k_mod in self._pairs():
            for pq, pk in zip(q_mod.parameters(), k_mod.parameters()):
                qs.append(pq.data)
                ks.append(pk)                              # the parameter itself: ema_update bumps its version counter
        from ...optim import ema_update
        ema_update(ks, qs, m)

    def _embed(self, seq, key: bool):
        e1, e2, e3, p1, p2, p3, head = ((self.encoder_k_1, self.encoder_k_2, self.encoder_k_3, self.proj_k_1, self.proj_k_2,
                                         self.proj_k_3, self.projector_k) if key else
                                        (self.encoder_1, self.encoder_2, self.encoder_3, self.proj1, self.proj2, self.proj3,
                                         self.projector))
        cat, (b, h, w) = decode_tokens(e1, e2, e3, p1, p2, p3, seq)
        proj = head.forward_tokens(cat, (b, h, w), LCAT)
        pred = F.normalize(proj.float(), dim=1)
        return H.from_tokens(pred, b, h, w)

    def forward(self, seq_1, seq_2, seq_3, seq_4, seq_5, seq_6):
        # (the num_batches_tracked increments of the 8 encoder passes - 240 one-element add kernels - are applied by one
        # foreach add on exit; the accumulators of all passes come out of one zero-filled block)
        hip.arena_reset(seq_1.device)
        seqs = (seq_1, seq_2, seq_3, seq_4, seq_5, seq_6)
        with H.deferred_bn_counters():
            if _batched_views_ok(self):
                # The 2 query and the 6 key passes as ONE batch each (view-interleaved clips, per-view BatchNorm statistics -
                # see headops.bn_views): 2x / 6x the rows per kernel, a quarter of the launches, and the query encoder's
                # parameters receive one gradient instead of two that autograd has to add up (255 add kernels per step)
                b = seq_1.shape[0]
                xq = torch.stack(seqs[:2], 1).reshape(2 * b, *seq_1.shape[1:])
                with H.bn_views(2, 2 * b):
                    pq = self._embed(xq, False)
                pred_1, pred_2 = pq[0::2], pq[1::2]
                with torch.no_grad():
                    self._momentum_update_key_encoder()
                    xk = torch.stack(seqs, 1).reshape(6 * b, *seq_1.shape[1:])
                    with H.bn_views(6, 6 * b):
                        pk = self._embed(xk, True)
                    keys = [pk[i::6] for i in range(6)]
            else:
                pred_1 = self._embed(seq_1, False)
                pred_2 = self._embed(seq_2, False)
                with torch.no_grad():
                    self._momentum_update_key_encoder()
                    keys = [self._embed(s, True) for s in seqs]
        return (pred_1, pred_2, *keys)


class ConsistencyLoss(nn.Module):
    """PixPro_swin_v5.py:565-597."""

    def __init__(self, args, input_resolution=(32, 56)):
        super().__init__()
        self.pixpro_pos_ratio = args.pixpro_pos_ratio
        self.pixpro = PixPro(args, input_resolution)
        if args.data == 'endo18':
            self.class_num = 12
        elif args.data == 'cata':
            self.class_num = int(num_class_table[args.tag])

    def forward(self, im_1, im_2, im_3, im_4, im_5, im_6, mask_1, mask_2, mask_3, mask_4, mask_5, mask_6):
        pred_1, pred_2, k1, k2, a1, a2, a3, n3 = self.pixpro(im_1, im_2, im_3, im_4, im_5, im_6)
        hh, ww = pred_1.shape[2:]
        m = [F.interpolate(x, size=[hh, ww], mode='nearest') for x in (mask_1, mask_2, mask_3, mask_4, mask_5, mask_6)]
        return regression_loss(pred_1, k2, a1, a2, a3, n3, m[0], m[1], m[2], m[3], m[4], m[5], self.class_num) \
            + regression_loss(pred_2, k1, a1, a2, a3, n3, m[1], m[0], m[2], m[3], m[4], m[5], self.class_num)
