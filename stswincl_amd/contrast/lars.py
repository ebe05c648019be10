"""LARS of the contrastive pre-training stage on the multi-tensor HIP kernel.

Mirror of pixcontrast_18/contrast/lars.py: ``add_weight_decay(model, weight_decay, skip_list)`` (:7-31) and
``LARS(optimizer, eps, trust_coef)`` (:33-152), used as ``LARS(torch.optim.SGD(add_weight_decay(model.pixpro, wd), lr, momentum))``
by main_pretrain_swinv5.py:37-47.  The wrapper keeps the reference's surface (param_groups / state / state_dict /
load_state_dict / zero_grad / add_param_group delegate to the wrapped optimizer, whose ``momentum_buffer`` state it updates), but
``step()`` runs every parameter group as TWO launches per 48 tensors (per-tensor norms, then the trust-ratio SGD-momentum update)
instead of 4 torch kernels + 2 host-synchronising norm reads per tensor.

Differences a caller can observe: ``p.grad`` is left untouched (the reference overwrites it with the decayed, scaled gradient
before the inner step; nothing reads it afterwards) and the wrapped optimizer must be SGD-like with dampening 0 and
nesterov off (anything else raises).
"""
from __future__ import annotations

import torch

from .. import hip
from ..optim import _mark_updated, group_clock

__all__ = ["LARS", "add_weight_decay"]


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """[{no-decay params (1-D or in skip_list): weight_decay 0, ignore True}, {the rest: weight_decay, ignore False}]."""
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        (no_decay if (param.dim() == 1 or name in skip_list) else decay).append(param)
    return [{"params": no_decay, "weight_decay": 0, "ignore": True},
            {"params": decay, "weight_decay": weight_decay, "ignore": False}]


class LARS:
    def __init__(self, optimizer, eps=1e-8, trust_coef=0.001):
        if eps < 0.0:
            raise ValueError("invalid epsilon value: , %f" % eps)
        if trust_coef < 0.0:
            raise ValueError("invalid trust coefficient: %f" % trust_coef)
        self.optim = optimizer
        self.eps = eps
        self.trust_coef = trust_coef

    def __getstate__(self):
        return (self.optim, {"eps": self.eps, "trust_coef": self.trust_coef})

    def __setstate__(self, state):
        self.optim, d = state
        self.eps, self.trust_coef = d["eps"], d["trust_coef"]

    def __repr__(self):
        return "%s(%r)" % (self.__class__.__name__, self.optim)

    @property
    def param_groups(self):
        return self.optim.param_groups

    @property
    def state(self):
        return self.optim.state

    def state_dict(self):
        sd = self.optim.state_dict()
        sd["param_groups"] = [{k: v for k, v in g.items() if k != "_clock"} for g in sd["param_groups"]]
        return sd

    def push_hyper(self):
        """Before a hipGraph replay of the step: the groups' current learning rates -> device (stswincl_amd.graph.GraphedStep)."""
        for group in self.optim.param_groups:
            if group.get("_clock") is not None:
                group["_clock"].push_lr(group["lr"])

    def load_state_dict(self, state_dict):
        self.optim.load_state_dict(state_dict)

    def zero_grad(self, set_to_none=True):
        self.optim.zero_grad(set_to_none=set_to_none)

    def add_param_group(self, param_group):
        self.optim.add_param_group(param_group)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:                    # (torch.optim re-enables grad for the closure: it runs forward + backward)
            with torch.enable_grad():
                loss = closure()
        for group in self.optim.param_groups:
            if group.get("dampening", 0) != 0 or group.get("nesterov", False) or group.get("maximize", False):
                raise hip.StswinHipError("fused LARS wraps plain SGD-momentum (dampening 0, nesterov off)")
            momentum = float(group.get("momentum", 0.0))
            wd = float(group["weight_decay"])
            ignore = group.get("ignore", None)              # set by add_weight_decay
            adaptive = ignore is not None and not ignore     # lars.py:129: groups without the key are never scaled
            first, later, touched = ([], [], []), ([], [], []), []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.optim.state[p]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if st.get("momentum_buffer") is None:
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    tgt = first
                else:
                    tgt = later
                tgt[0].append(p.data)
                tgt[1].append(g)
                tgt[2].append(st["momentum_buffer"])
                touched.append(p)
            for (ps, gs, ms), is_first in ((first, True), (later, False)):
                if ps:
                    # (the learning rate travels through device memory: a hipGraph replay of the step sees the scheduler's current value)
                    hip.multi_tensor_lars(ps, gs, ms, None, lr=float(group["lr"]), momentum=momentum,
                                          wd=wd, trust_coef=self.trust_coef, eps=self.eps, first=is_first, adaptive=adaptive,
                                          hyper=group_clock(group, ps[0].device).hyper)
            _mark_updated(touched)
        return loss
