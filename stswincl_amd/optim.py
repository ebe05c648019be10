"""Optimizers of the reference's training scripts on the multi-tensor HIP kernel (csrc/optim.hip).

``FusedAdam`` == torch.optim.Adam(params, lr, betas, eps, weight_decay) as used by seg18/train_swin.py:122;
``FusedSGD``  == torch.optim.SGD(params, lr, momentum, weight_decay) with per-group lr / weight_decay
                 (train_CL_ft_mswin_sgd_minput.py:147-162); the LARS wrapper of the contrastive stage is
                 stswincl_amd/contrast/lars.py (``make_contrast_optimizer`` builds main_pretrain_swinv5.py:37-47's stack);
``ema_update``== PixPro._momentum_update_key_encoder (PixPro_swin_v5.py:258-289) in ~8 launches instead of ~370x2.
"""
from __future__ import annotations

import math
from typing import Iterable, Sequence

import torch

import os

from . import hip

_EAGER_REPACK = os.environ.get("STSWIN_LAZY_REPACK") != "1"       # (A/B switch: per-weight re-packing at the next use)


def _mark_updated(params: Sequence[torch.Tensor]) -> None:
    """The multi-tensor kernel writes through raw pointers, which autograd's version counters do not see - and the bf16
    weight cache of the GEMM path (ops.wcast) is keyed on `_version`, exactly like anything else that memoises on a
    parameter.  Bump the counters of the tensors just written (no kernel launch)."""
    if not params:
        return
    setter = getattr(torch._C._autograd, "_unsafe_set_version_counter", None)
    if setter is not None:
        setter(list(params), [int(p._version) + 1 for p in params])
    else:                                  # older torch: an in-place no-op through the dispatcher
        torch._foreach_add_(list(params), 0)
    if _EAGER_REPACK:                      # ... and re-make their cached GEMM operands now, batched (two launches instead of ~90)
        from . import ops
        ops.repack(params)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:                    # (torch.optim re-enables grad for the closure: it runs forward + backward)
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            by_step = {}          # torch.optim.Adam keeps the step count PER PARAMETER (bias corrections differ when a branch
            touched = []          # had no gradient on some steps, or a parameter was unfrozen later): one launch per count
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if not isinstance(st["step"], int):          # a state loaded from torch.optim.Adam holds tensor steps: one host
                    st["step"] = int(st["step"])             # read, here, before any capture (see load_state_dict)
                st["step"] += 1
                ps, gs, ms, vs = by_step.setdefault(st["step"], ([], [], [], []))
                ps.append(p.data)
                touched.append(p)
                gs.append(p.grad.contiguous() if not p.grad.is_contiguous() else p.grad)
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
            for step, (ps, gs, ms, vs) in by_step.items():
                hip.multi_tensor(0, ps, gs, ms, vs, lr=group["lr"], b1=b1, b2=b2, eps=group["eps"], wd=group["weight_decay"],
                                 c1=1.0 - b1 ** step, c2=math.sqrt(1.0 - b2 ** step))
            _mark_updated(touched)
        return loss


    def load_state_dict(self, state_dict):
        """torch.optim.Adam checkpoints keep `step` as a (possibly GPU) tensor per parameter; the fused kernel takes the bias
        corrections as host scalars, so the counts become Python ints here - once, outside any hipGraph capture - instead of
        forcing a device-to-host read inside step()."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if "step" in st and not isinstance(st["step"], int):
                st["step"] = int(st["step"])


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:                    # (torch.optim re-enables grad for the closure: it runs forward + backward)
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            first, later, touched = ([], [], []), ([], [], []), []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                g = p.grad.contiguous() if not p.grad.is_contiguous() else p.grad
                if st.get("momentum_buffer") is None:      # (a loaded state may hold None: first step)
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    tgt = first
                else:
                    tgt = later
                tgt[0].append(p.data)
                touched.append(p)
                tgt[1].append(g)
                tgt[2].append(st["momentum_buffer"])
            for (ps, gs, ms), c1 in ((first, 1.0), (later, 0.0)):
                if ps:
                    hip.multi_tensor(1, ps, gs, ms, None, lr=group["lr"], b1=group["momentum"], wd=group["weight_decay"], c1=c1)
            _mark_updated(touched)
        return loss


@torch.no_grad()
def ema_update(keys: Sequence[torch.Tensor], queries: Sequence[torch.Tensor], momentum: float) -> None:
    """key <- key * momentum + query * (1 - momentum) for every pair.  Pass the key PARAMETERS (not their .data aliases):
    their version counters are bumped so that caches keyed on them see the update."""
    hip.multi_tensor(2, [k.data for k in keys], [q.data for q in queries], b1=momentum)
    _mark_updated(list(keys))


def make_contrast_optimizer(params, batch_size: int, base_learning_rate: float = 1.0, momentum: float = 0.9,
                            weight_decay: float = 1e-5, optimizer: str = "lars"):
    """The optimizer stack of main_pretrain_swinv5.py:32-47 on the fused kernels: lr = global batch / 256 * base lr;
    'lars': add_weight_decay groups (1-D parameters: no decay, no trust ratio) + SGD momentum under LARS; 'sgd': plain
    SGD momentum with weight decay.  -> (optimizer, short name for reports)."""
    from .contrast.lars import LARS
    params = list(params)
    lr = batch_size / 256.0 * base_learning_rate
    if optimizer == "sgd":
        return FusedSGD(params, lr, momentum=momentum, weight_decay=weight_decay), "SGD"
    groups = [{"params": [p for p in params if p.dim() == 1], "weight_decay": 0, "ignore": True},
              {"params": [p for p in params if p.dim() != 1], "weight_decay": weight_decay, "ignore": False}]
    return LARS(FusedSGD(groups, lr, momentum=momentum)), "LARS(SGD)"
