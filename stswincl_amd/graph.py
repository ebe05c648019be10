"""hipGraph replay of a whole training step as a TRAINING mode (round 6), not only a measurement mode.

Every kernel of libstswin_hip is launched on the caller's stream with caller-owned workspaces and no host synchronisation, so a step
(forward + loss + backward + optimizer) captures into one graph.  What used to make a replay differ from an eager step were host
scalars baked into kernel arguments: Adam's step count / bias corrections, the learning rate of SGD / LARS, the key-encoder momentum
schedule of the contrastive stage.  They now live in device memory (stswincl_amd.optim._Clock / EmaSchedule, csrc/optim.hip
stswin_optim_tick): counters advance inside the graph, and values the HOST changes between steps (a scheduler's learning rate) are
handed over by a stream-ordered fill in front of the replay (`push_hyper`).  N replays == N eager steps, bit for bit
(tests/test_hip_graph_training.py).

    step = GraphedStep(step_fn, [opt], zero_grad=lambda: opt.zero_grad(set_to_none=True))
    for batch in loader:
        static_x.copy_(batch.x, non_blocking=True); static_y.copy_(batch.y, non_blocking=True)     # stream-ordered, into the captured buffers
        scheduler.step()                                                                           # host: group['lr'] changes
        loss = step()                                                                              # push_hyper + one graph launch

Reference loops this replaces: seg18/train_swin.py:151-173, pixcontrast_18/main_pretrain_swinv5.py:113-153.
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch

from . import hip


class GraphedStep:
    """Capture `step_fn` (which must run opt.zero_grad / forward / backward / opt.step on static input buffers and return the loss
    tensor) after `warmup` eager executions; calling the object replays it.  The warm-up executions ARE training steps (they update
    the parameters); the capture itself executes nothing."""

    def __init__(self, step_fn: Callable[[], torch.Tensor], optimizers: Iterable = (), zero_grad: Optional[Callable[[], None]] = None,
                 warmup: int = 2, before_step: Optional[Callable[[int], None]] = None):
        """before_step(i): host-side work of step i that stays OUTSIDE the graph - copy the batch into the static input buffers,
        advance a learning-rate scheduler; called before every warm-up step and every replay (i counts from 0)."""
        self.optimizers = list(optimizers)
        self.before_step = before_step
        self.steps_run = 0
        self.warmup_losses = []
        side = self.stream = torch.cuda.Stream()         # (eager steps run later should use it too: the parameters' AccumulateGrad nodes
        side.wait_stream(torch.cuda.current_stream())     #  are bound to the stream of their first backward)
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):           # (at least one: optimizer state and device clocks must exist before the capture)
                if before_step is not None:
                    before_step(self.steps_run)
                self.warmup_losses.append(step_fn().detach().clone())
                self.steps_run += 1
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if zero_grad is not None:
            zero_grad()
        hip.note_capture()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = step_fn()
        # the capture ran the host side of one step without executing it: host mirrors of device counters are one ahead until re-read
        for o in self.optimizers:
            if hasattr(o, "_replayed"):
                o._replayed = True

    def __call__(self) -> torch.Tensor:
        if self.before_step is not None:
            self.before_step(self.steps_run)
        for o in self.optimizers:
            push = getattr(o, "push_hyper", None)
            if push is not None:
                push()
        self.graph.replay()
        self.steps_run += 1
        return self.loss
