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


class ContrastBankFn(torch.autograd.Function):
    """(queries, labels, key bank, bank labels) -> pos, neg fp32 [M][groups] (+ rowmax, lse [M]): the masked similarity sums over
    label-equal / label-different visible bank rows, one launch for every key map and both loss directions
    (csrc/contrast.hip, stswin_contrast_bank_fwd).  The bank is no-grad (keys come from the momentum encoders,
    PixPro_swin_v5.py:366).  Backward to the queries, all HIP: per-class sums of the bank rows (exact fp32 adds, so an empty set
    gives an exactly-zero gradient like the reference's masked products) and one combination pass:
    dq_m = sum_g dpos_mg Kcls_g[l_m] + dneg_mg (Ktot_g - Kcls_g[l_m])."""

    @staticmethod
    def forward(ctx, q, lq, bank, lb, cnt, cfg):
        q_sets, q_block, bank_block, gmap, ncls, inv_tau, want_lse = cfg
        Q = q.detach().to(bank.dtype).contiguous()
        pos, tot, rmax, lse = hip.contrast_bank_fwd(Q, lq, bank, lb, q_sets=q_sets, q_block=q_block, bank_block=bank_block,
                                                    gmap=gmap, inv_tau=inv_tau, want_lse=want_lse)
        ctx.cfg = (q_sets, q_block, bank_block, gmap, ncls, q.dtype)
        ctx.save_for_backward(lq, bank, lb, cnt)
        if not want_lse:
            rmax = lse = pos.new_zeros(())
        ctx.mark_non_differentiable(rmax, lse)
        return pos, tot - pos, rmax, lse

    @staticmethod
    def backward(ctx, dpos, dneg, _dm, _dl):
        lq, bank, lb, cnt = ctx.saved_tensors
        q_sets, q_block, bank_block, gmap, ncls, in_dtype = ctx.cfg
        ksum = hip.contrast_class_sums(bank, lb, bank_block, ncls)
        dq = hip.contrast_bank_dq(dpos.float(), dneg.float(), cnt, lq, ksum, q_sets=q_sets, q_block=q_block, seg=bank.shape[1],
                                  bank_block=bank_block, gmap=gmap)
        return dq.to(in_dtype), None, None, None, None, None


_COUNT_INDEX: dict = {}


def _count_index(M, q_sets, q_block, nb, gmap, device):
    """(map index [M][groups], bank block [M][1]) of every query row; cached per geometry (built on the host once: an
    H2D copy per step would also be illegal inside a hipGraph capture)."""
    key = (M, q_sets, q_block, nb, tuple(map(tuple, gmap)), str(device))
    hit = _COUNT_INDEX.get(key)
    if hit is None:
        rows = torch.arange(M)
        per_set = M // q_sets
        blk = ((rows % per_set) // q_block) if nb > 1 else torch.zeros_like(rows)
        gm = torch.tensor(gmap)[rows // per_set]
        hit = (gm.to(device), blk[:, None].to(device))
        _COUNT_INDEX[key] = hit
    return hit


def _label_counts(lq, lb, gmap, q_sets, q_block, bank_block, class_num):
    """cnt[m][g] = number of visible bank rows of group g whose label equals lq[m] (the |posMask| row sums of
    PixPro_swin_v5.py:116-118), from per-block label histograms: O(rows) integer work instead of HW x HW masks."""
    maps, seg = lb.shape
    nb = seg // bank_block
    hist = F.one_hot(lb.long().clamp(0, class_num - 1), class_num).view(maps, nb, bank_block, class_num).sum(2)   # [maps][nb][cls]
    gm, blk = _count_index(lq.shape[0], q_sets, q_block, nb, gmap, lq.device)
    return hist[gm, blk, lq.long().clamp(0, class_num - 1)[:, None]].float()             # [M][groups]


def bank_contrast_loss(q_tok, lq, bank, lb, gmap, q_block, bank_block, class_num, inv_tau=1.0, want_lse=False):
    """The loss of PixPro_swin_v5.py:71-129 for `len(gmap)` query sets against a key bank (see stswin_contrast_bank_fwd):
    P = sum_g pos / (sum_g |pos| + 1e-6), N = sum_g neg_g / (|neg_g| + 1e-6), loss = sum over the query sets of
    -mean(log(e^P / (e^P + e^N) + 1e-6)).  -> (loss, rowmax, lse)."""
    q_sets = len(gmap)
    cnt = _label_counts(lq, lb, gmap, q_sets, q_block, bank_block, class_num)
    pos, neg, rmax, lse = ContrastBankFn.apply(q_tok, lq, bank, lb, cnt, (q_sets, q_block, bank_block, tuple(map(tuple, gmap)),
                                                                          class_num, float(inv_tau), bool(want_lse)))
    P = pos.sum(-1) / (cnt.sum(-1) + 1e-6)
    Nn = (neg / ((bank_block - cnt) + 1e-6)).sum(-1)
    pe, ne = torch.exp(P), torch.exp(Nn)
    term = -torch.log(pe / (pe + ne) + 1e-6)
    return term.view(q_sets, -1).mean(1).sum(), rmax, lse


def _tokens_and_labels(feats, labels, dt):
    n, c, h, w = feats[0].shape
    toks = torch.stack([H.to_tokens(t).detach().to(dt) for t in feats], 0).contiguous()          # [maps][N*HW][C]
    labs = torch.stack([l.reshape(n * h * w).to(torch.int32) for l in labels], 0).contiguous()   # [maps][N*HW]
    return toks, labs


def regression_loss(q, k, adj1, adj2, adj3, neg3, label_patch1, label_patch2, label_adj1, label_adj2, label_adj3,
                    label_neg3, class_num):
    """PixPro_swin_v5.py:71-129.  P = sum_j sum_pos / (sum_j |pos| + 1e-6); N = sum_j sum_neg_j / (|neg_j| + 1e-6);
    loss = -mean(log(e^P / (e^P + e^N) + 1e-6)).  No temperature, no softmax over keys."""
    n, c, h, w = q.shape
    HW = h * w
    dt = compute_dtype(q)
    bank, lb = _tokens_and_labels((k, adj1, adj2, adj3, neg3), (label_patch2, label_adj1, label_adj2, label_adj3, label_neg3), dt)
    lq = label_patch1.reshape(n * HW).to(torch.int32).contiguous()
    loss, _, _ = bank_contrast_loss(H.to_tokens(q), lq, bank, lb, [[0, 1, 2, 3, 4]], HW, HW, class_num)
    return loss


def gather_bank(bank: torch.Tensor, lb: torch.Tensor, group=None):
    """Inter-video key bank: all-gather every key map (and its labels) over the ranks into [maps][world * rows][C] - one
    RCCL all-gather per map straight into its slice (no re-layout copy); keys are no-grad, so there is no backward collective.
    The reference's (unused) dist_collect, pixcontrast_18/contrast/util.py:47-58."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bank, lb
    world = dist.get_world_size(group)
    maps, rows, C = bank.shape
    out = torch.empty(maps, world * rows, C, dtype=bank.dtype, device=bank.device)
    lo = torch.empty(maps, world * rows, dtype=lb.dtype, device=lb.device)
    for m in range(maps):
        dist.all_gather_into_tensor(out[m], bank[m].contiguous(), group=group)
        dist.all_gather_into_tensor(lo[m], lb[m].contiguous(), group=group)
    return out, lo


def consistency_pair_loss(pred_1, pred_2, k1, k2, a1, a2, a3, n3, m, class_num, bank_mode="sample", inv_tau=1.0, want_lse=False):
    """Both regression_loss calls of ConsistencyLoss.forward (PixPro_swin_v5.py:594-595) as ONE similarity launch.
    bank_mode 'sample': the reference - a query pixel sees the five key maps of its own sample; 'batch': the key maps of every
    sample of this rank (inter-video); 'world': of every sample of every rank (RCCL all-gather of the keys, gather_bank)."""
    n, c, h, w = pred_1.shape
    HW = h * w
    dt = compute_dtype(pred_1)
    bank, lb = _tokens_and_labels((k1, k2, a1, a2, a3, n3), m, dt)
    q_tok = torch.cat([H.to_tokens(pred_1), H.to_tokens(pred_2)], 0)
    lq = torch.cat([lb[0], lb[1]], 0)
    gmap = [[1, 2, 3, 4, 5], [0, 2, 3, 4, 5]]
    if bank_mode == "sample":
        q_block = bank_block = HW
    else:
        if bank_mode == "world":
            bank, lb = gather_bank(bank, lb)
        elif bank_mode != "batch":
            raise ValueError(f"bank_mode {bank_mode!r}")
        q_block, bank_block = n * HW, bank.shape[1]
    return bank_contrast_loss(q_tok, lq, bank, lb, gmap, q_block, bank_block, class_num, inv_tau, want_lse)


class PairLossFn(torch.autograd.Function):
    """Round 5: the whole tail of ConsistencyLoss.forward behind the projector as ONE autograd node of HIP launches -
    F.normalize of the query embeddings, both regression_loss calls (PixPro_swin_v5.py:594-595) and every reduction between them:
    proj_q [2 b HW][C] (clip-major token rows of the batched query pass, pre-normalisation) -> loss.  Forward: rownorm_scatter
    (normalise + de-interleave the two views), label counts, the bank similarity kernel, the loss kernel; backward: loss derivative,
    per-class key sums, query gradient, normalisation backward.  The reference formulation ran as ~60 torch elementwise / reduce /
    copy launches forward and as many backward."""

    @staticmethod
    def forward(ctx, proj_q, lq, bank, lb, cfg):
        b, HW, q_block, bank_block, class_num, inv_tau, want_lse = cfg
        gmap = ((1, 2, 3, 4, 5), (0, 2, 3, 4, 5))
        X = proj_q.detach()
        if X.stride(1) != 1:
            X = X.contiguous()
        q_tok = torch.empty(2 * b * HW, X.shape[1], dtype=X.dtype, device=X.device)
        inv = hip.rownorm_scatter(X, q_tok, 2, HW, b, want_inv=True)
        # lq: THIS rank's labels of the two query views (rows of the LOCAL lb[:2], taken by the caller before any all-gather of
        # the bank: a gathered lb is rank-major per map and 'world' times as long as the query matrix)
        if lq.numel() != q_tok.shape[0]:
            raise ValueError(f"PairLossFn: {lq.numel()} query labels for {q_tok.shape[0]} query rows")
        cnt = hip.label_counts(lq, lb, q_sets=2, q_block=q_block, bank_block=bank_block, ncls=class_num, gmap=gmap)
        # (unit_rows: both operands were normalised by rownorm_scatter - the log-sum-exp takes the fixed-reference form)
        pos, tot, rmax, lse = hip.contrast_bank_fwd(q_tok, lq, bank, lb, q_sets=2, q_block=q_block, bank_block=bank_block, gmap=gmap,
                                                    inv_tau=inv_tau, want_lse=want_lse, unit_rows=True)
        loss = hip.pair_loss(pos, tot, cnt, 2, bank_block)
        ctx.cfg = (b, HW, q_block, bank_block, class_num, gmap)
        ctx.save_for_backward(X, inv, lq, bank, lb, cnt, pos, tot)
        if not want_lse:
            rmax = lse = pos.new_zeros(())
        ctx.mark_non_differentiable(rmax, lse)
        return loss.view(()), rmax, lse

    @staticmethod
    def backward(ctx, dloss, _dm, _dl):
        X, inv, lq, bank, lb, cnt, pos, tot = ctx.saved_tensors
        b, HW, q_block, bank_block, class_num, gmap = ctx.cfg
        dpos, dneg = hip.pair_loss_bwd(pos, tot, cnt, dloss.detach().float().contiguous(), 2, bank_block)
        ksum = hip.contrast_class_sums(bank, lb, bank_block, class_num)
        dq = hip.contrast_bank_dq(dpos, dneg, cnt, lq, ksum, q_sets=2, q_block=q_block, seg=bank.shape[1], bank_block=bank_block, gmap=gmap)
        return hip.rownorm_scatter_bwd(X, inv, dq, 2, HW, b), None, None, None, None


def pair_loss_tokens(proj_q, bank, lb, b, HW, class_num, bank_mode="sample", inv_tau=1.0, want_lse=False):
    """consistency_pair_loss on the token operands the batched encoder passes leave behind: proj_q = projector output of the batched
    query pass (clip-major rows, before F.normalize), bank [6][b HW][C] = normalised key embeddings (view-major), lb int32 [6][b HW]."""
    lq = lb[:2].reshape(-1)           # the two query views' labels of this rank (a view of the local lb; BEFORE the gather)
    if bank_mode == "sample":
        q_block = bank_block = HW
    else:
        if bank_mode == "world":
            bank, lb = gather_bank(bank, lb)
        elif bank_mode != "batch":
            raise ValueError(f"bank_mode {bank_mode!r}")
        q_block, bank_block = b * HW, bank.shape[1]
    return PairLossFn.apply(proj_q, lq, bank, lb, (b, HW, q_block, bank_block, class_num, float(inv_tau), bool(want_lse)))


def Proj_Head(in_dim=400, inner_dim=512, out_dim=256):
    return MLP2d(in_dim, inner_dim, out_dim)


def Pred_Head(in_dim=256, inner_dim=4096, out_dim=256):
    return MLP2d(in_dim, inner_dim, out_dim)


LCAT = H.Layout.concat([H.Layout.dense(48)] * 3 + [H.Layout.dense(256)])


def _batched_views_ok(model) -> bool:
    """View batching (interleaved-group BatchNorm kernels; SyncBatchNorm gathers the per-view statistics over the ranks in the same
    one collective per layer): train mode, and not switched off (STSWIN_SEQUENTIAL_VIEWS=1: A/B runs)."""
    return os.environ.get("STSWIN_SEQUENTIAL_VIEWS") != "1" and model.training


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
            from ...utils.LoadModel import _torch_load
            sd = _torch_load(pre, "cpu")
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
        self._ema = None                  # optim.EmaSchedule: k and the momentum of the step in device memory (made at the first update)
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

    @property
    def k(self) -> int:
        """Key-encoder updates done so far (PixPro_swin_v5.py:246,262).  Once the schedule lives on the device this is its host mirror:
        exact in eager loops; after hipGraph replays of the step call `sync_k()` (one device read) before relying on it."""
        return self._ema.k if self._ema is not None else self._k

    @k.setter
    def k(self, value) -> None:
        self._k = int(value)
        self._ema = None                   # re-made from _k at the next update

    def sync_k(self) -> int:
        if self._ema is not None:
            self._k = self._ema.sync()
        return self._k

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """k <- k m + q (1-m), m = 1 - (1-m0)(cos(pi k/K)+1)/2 (PixPro_swin_v5.py:258-289), on the multi-tensor HIP kernel.  The step
        counter k and the momentum m live in device memory (optim.EmaSchedule: one one-thread launch evaluates the reference's
        expression in double precision and advances k), so a hipGraph replay of the training step walks the schedule like eager
        steps do."""
        from ...optim import EmaSchedule, ema_update
        qs, ks = [], []
        for q_mod, k_mod in self._pairs():
            for pq, pk in zip(q_mod.parameters(), k_mod.parameters()):
                qs.append(pq.data)
                ks.append(pk)                              # the parameter itself: ema_update bumps its version counter
        if self._ema is None or self._ema.clock.counter.device != ks[0].device or self._ema.K != float(self.K) \
                or self._ema.m0 != float(self.pixpro_momentum):
            if torch.cuda.is_current_stream_capturing():
                raise hip.StswinHipError("PixPro: first key-encoder update inside a hipGraph capture; run one eager step first")
            self._ema = EmaSchedule(ks[0].device, self.pixpro_momentum, self.K, self._k)
        ema_update(ks, qs, 0.0, hyper=self._ema.tick())

    def _embed(self, seq, key: bool, tokens: bool = False):
        e1, e2, e3, p1, p2, p3, head = ((self.encoder_k_1, self.encoder_k_2, self.encoder_k_3, self.proj_k_1, self.proj_k_2,
                                         self.proj_k_3, self.projector_k) if key else
                                        (self.encoder_1, self.encoder_2, self.encoder_3, self.proj1, self.proj2, self.proj3,
                                         self.projector))
        cat, (b, h, w) = decode_tokens(e1, e2, e3, p1, p2, p3, seq)
        proj = head.forward_tokens(cat, (b, h, w), LCAT)
        if tokens:
            return proj, (b, h, w)
        pred = F.normalize(proj.float(), dim=1)
        return H.from_tokens(pred, b, h, w)

    def forward_tokens(self, seqs):
        """The eight encoder passes for ConsistencyLoss (round 5): -> (proj_q, bank, (b, h, w)).  proj_q [2 b h w][C]: the projector
        output of the batched query pass (clip-major rows: clip = sample * 2 + view; carries the gradient; NOT yet normalised - the pair
        loss normalises it in its first kernel).  bank [6][b h w][C]: the six key views' normalised embeddings, view-major - written
        in place by one normalise-and-scatter kernel instead of normalize -> NCHW -> six strided slices -> tokens -> stack."""
        H.refuse_replica(self)
        hip.arena_reset(seqs[0].device)
        b = seqs[0].shape[0]
        with H.deferred_bn_counters():
            xq = torch.stack(seqs[:2], 1).reshape(2 * b, *seqs[0].shape[1:])
            with H.bn_views(2, 2 * b):
                proj_q, (_, h, w) = self._embed(xq, False, tokens=True)
            with torch.no_grad():
                self._momentum_update_key_encoder()
                xk = torch.stack(seqs, 1).reshape(6 * b, *seqs[0].shape[1:])
                with H.bn_views(6, 6 * b):
                    proj_k, _ = self._embed(xk, True, tokens=True)
                C = proj_k.shape[1]
                bank = torch.empty(6, b * h * w, C, dtype=proj_k.dtype, device=proj_k.device)
                hip.rownorm_scatter(proj_k if proj_k.stride(1) == 1 else proj_k.contiguous(), bank.view(6 * b * h * w, C), 6, h * w, b)
        return proj_q, bank, (b, h, w)

    def forward(self, seq_1, seq_2, seq_3, seq_4, seq_5, seq_6):
        # (the num_batches_tracked increments of the 8 encoder passes - 240 one-element add kernels - are applied by one
        # foreach add on exit; the accumulators of all passes come out of one zero-filled block)
        H.refuse_replica(self)
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
        # extension (not in the reference's option list; default = the reference): 'sample' | 'batch' | 'world' key bank,
        # see consistency_pair_loss.  pixpro_bank_stats additionally returns the row-max / log-sum-exp of the scores / tau.
        self.bank_mode = getattr(args, "pixpro_bank", "sample")
        self.bank_inv_tau = 1.0 / float(getattr(args, "pixpro_bank_tau", 1.0))
        self.bank_stats = bool(getattr(args, "pixpro_bank_stats", False))
        self.last_rowmax = self.last_lse = None

    def forward(self, im_1, im_2, im_3, im_4, im_5, im_6, mask_1, mask_2, mask_3, mask_4, mask_5, mask_6):
        if _batched_views_ok(self.pixpro) and os.environ.get("STSWIN_CONTRAST_TORCH_GLUE") != "1":
            # token path (round 5): the embeddings never leave their token matrices and the glue between the encoders and the loss is
            # six HIP launches (STSWIN_CONTRAST_TORCH_GLUE=1: the torch formulation below, for A/B runs and the parity test)
            proj_q, bank, (b, hh, ww) = self.pixpro.forward_tokens((im_1, im_2, im_3, im_4, im_5, im_6))
            lb = hip.labels_resize((mask_1, mask_2, mask_3, mask_4, mask_5, mask_6), hh, ww)
            loss, self.last_rowmax, self.last_lse = pair_loss_tokens(proj_q, bank, lb, b, hh * ww, self.class_num, self.bank_mode,
                                                                     self.bank_inv_tau, self.bank_stats)
            return loss
        pred_1, pred_2, k1, k2, a1, a2, a3, n3 = self.pixpro(im_1, im_2, im_3, im_4, im_5, im_6)
        hh, ww = pred_1.shape[2:]
        m = [F.interpolate(x, size=[hh, ww], mode='nearest') for x in (mask_1, mask_2, mask_3, mask_4, mask_5, mask_6)]
        # regression_loss(pred_1 | k2, a1, a2, a3, n3) + regression_loss(pred_2 | k1, a1, a2, a3, n3)  (:594-595), one launch
        loss, self.last_rowmax, self.last_lse = consistency_pair_loss(pred_1, pred_2, k1, k2, a1, a2, a3, n3, m, self.class_num,
                                                                      self.bank_mode, self.bank_inv_tau, self.bank_stats)
        return loss
