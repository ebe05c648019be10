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


class _Clock:
    """Device-resident step state shared by the parameters of one Adam launch group: `counter` int32 [1] (steps taken) and `hyper`
    fp32 [4] = {lr, 1 - b1^t, sqrt(1 - b2^t), -} (csrc/optim.hip, stswin_optim_tick).  The update kernels read their step-dependent
    scalars from `hyper`, so a hipGraph replay of the step advances the bias corrections exactly like eager steps do; the host
    mirror `step` is what state_dict() reports and is re-read from the device after replays (sync())."""

    def __init__(self, device, step: int = 0):
        self.counter = torch.full((1,), int(step), dtype=torch.int32, device=device)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=device)
        self.step = int(step)
        self.lr = None                       # the value hyper[0] holds
        self.members = []                    # optimizer state dicts of the parameters that took the last step on this clock

    def push_lr(self, lr: float) -> None:
        """Stream-ordered fill of hyper[0] when the host's learning rate differs from what the device holds (schedulers); a no-op
        otherwise - in particular inside a capture whose warm-up steps already ran with this rate."""
        lr = float(lr)
        if self.lr != lr:
            self.hyper[0:1].fill_(lr)
            self.lr = lr

    def sync(self) -> int:
        self.step = int(self.counter.item())
        return self.step


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._replayed = False
        self._clock_groups = []

    def _clocks(self):
        seen = {}
        for st in self.state.values():
            c = st.get("_clock")
            if c is not None:
                seen[id(c)] = c
        return list(seen.values())

    def push_hyper(self) -> None:
        """Before a graph replay: hand the groups' current learning rates to the device (stswincl_amd.graph.GraphedStep calls this)."""
        self._replayed = True
        for clock, group in self._clock_groups:         # (clock, group) pairs of the last step() - the one that was captured
            clock.push_lr(group["lr"])

    def sync_steps(self) -> None:
        """Host step counts <- device counters (after graph replays the host mirrors are stale)."""
        for c in self._clocks():
            c.sync()
            for st in c.members:            # (a parameter that shares the clock but sat the captured step out keeps its own count)
                st["step"] = c.step
        self._replayed = False

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:                    # (torch.optim re-enables grad for the closure: it runs forward + backward)
            with torch.enable_grad():
                loss = closure()
        capturing = torch.cuda.is_current_stream_capturing()
        if self._replayed and not capturing:       # eager steps after graph replays: the host mirrors first
            self.sync_steps()
        self._clock_groups = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            by_clock = {}         # torch.optim.Adam keeps the step count PER PARAMETER (bias corrections differ when a branch
            touched = []          # had no gradient on some steps, or a parameter was unfrozen later): one launch per count.
            fresh = {}            # Parameters at the same count share a device clock.
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if not isinstance(st["step"], int):          # a state loaded from torch.optim.Adam holds tensor steps: one host
                    st["step"] = int(st["step"])             # read, here, before any capture (see load_state_dict)
                clock = st.get("_clock")
                if clock is not None and clock.step != st["step"]:
                    clock = None                             # it sat out steps the clock's other parameters took: its own count, its own clock
                if clock is None:                            # first gradient (or a loaded state): join the clock of this count
                    if capturing:
                        raise hip.StswinHipError("FusedAdam: a parameter got its first gradient inside a hipGraph capture; run the "
                                                 "step eagerly once before capturing it")
                    key = (st["step"], p.device)
                    clock = fresh.get(key)
                    if clock is None:
                        clock = fresh[key] = next((c for (c, *_r) in by_clock.values() if c.step == st["step"] and c.counter.device == p.device),
                                                  None) or _Clock(p.device, st["step"])
                    st["_clock"] = clock
                _c, ps, gs, ms, vs, sts = by_clock.setdefault(id(clock), (clock, [], [], [], [], []))
                ps.append(p.data)
                touched.append(p)
                gs.append(p.grad.contiguous() if not p.grad.is_contiguous() else p.grad)
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
                sts.append(st)
            for clock, ps, gs, ms, vs, sts in by_clock.values():
                self._clock_groups.append((clock, group))
                clock.push_lr(group["lr"])
                hip.optim_tick(0, clock.counter, clock.hyper, float(b1), float(b2))
                clock.step += 1
                clock.members = sts
                for st in sts:
                    st["step"] = clock.step
                hip.multi_tensor(0, ps, gs, ms, vs, b1=b1, b2=b2, eps=group["eps"], wd=group["weight_decay"], hyper=clock.hyper)
            _mark_updated(touched)
        return loss

    def state_dict(self):
        """(the device clocks stay out of the checkpoint: `step` per parameter is what torch.optim.Adam writes, too)"""
        if self._replayed:
            self.sync_steps()
        sd = super().state_dict()
        sd["state"] = {k: {kk: vv for kk, vv in v.items() if kk != "_clock"} for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        """torch.optim.Adam checkpoints keep `step` as a (possibly GPU) tensor per parameter; the counts become Python ints here -
        once, outside any hipGraph capture - and the device clocks are re-made from them at the next step()."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            st.pop("_clock", None)
            if "step" in st and not isinstance(st["step"], int):
                st["step"] = int(st["step"])


def group_clock(group, device) -> _Clock:
    """The device-resident learning rate of a parameter group (SGD / LARS: lr is their only step-dependent scalar)."""
    c = group.get("_clock")
    if c is None or c.hyper.device != device:
        c = group["_clock"] = _Clock(device)
    c.push_lr(group["lr"])
    return c


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    def push_hyper(self) -> None:
        for group in self.param_groups:
            if group.get("_clock") is not None:
                group["_clock"].push_lr(group["lr"])

    def state_dict(self):
        sd = super().state_dict()
        sd["param_groups"] = [{k: v for k, v in g.items() if k != "_clock"} for g in sd["param_groups"]]
        return sd

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
                    hip.multi_tensor(1, ps, gs, ms, None, b1=group["momentum"], wd=group["weight_decay"], c1=c1,
                                     hyper=group_clock(group, ps[0].device).hyper)
            _mark_updated(touched)
        return loss


@torch.no_grad()
def ema_update(keys: Sequence[torch.Tensor], queries: Sequence[torch.Tensor], momentum: float, hyper=None) -> None:
    """key <- key * momentum + query * (1 - momentum) for every pair.  Pass the key PARAMETERS (not their .data aliases):
    their version counters are bumped so that caches keyed on them see the update.  hyper (device fp32 [4]): the momentum is read
    from hyper[3] (EmaSchedule) instead of the argument."""
    hip.multi_tensor(2, [k.data for k in keys], [q.data for q in queries], b1=momentum, hyper=hyper)
    _mark_updated(list(keys))


class EmaSchedule:
    """The key-encoder momentum schedule of PixPro_swin_v5.py:258-262 - m = 1 - (1 - m0)(cos(pi k / K) + 1) / 2, k += 1 per update -
    with k and m in device memory (stswin_optim_tick kind 1), so that eager steps and hipGraph replays of the step walk the same
    schedule.  `k` (host mirror) is re-read from the device by sync()."""

    def __init__(self, device, base_momentum: float, K: int, k: int = 0):
        self.clock = _Clock(device, k)
        self.m0, self.K = float(base_momentum), float(K)

    def tick(self) -> torch.Tensor:
        hip.optim_tick(1, self.clock.counter, self.clock.hyper, self.m0, self.K)
        self.clock.step += 1
        return self.clock.hyper

    @property
    def k(self) -> int:
        return self.clock.step

    def sync(self) -> int:
        return self.clock.sync()


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
