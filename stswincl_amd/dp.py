"""Data-parallel gradient averaging for the seg / contrastive training steps: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference uses nn.DataParallel (seg18/train_swin.py:131-135) or DDP (pixcontrast_18/main_pretrain_swinv5.py:54);
both reduce to: every rank holds the full model, processes its shard of clips, and gradients are averaged.
Clips are independent, so the only collective on the data path is this all-reduce.  Buckets are persistent flat fp32
buffers (optionally compressed to bf16 for the wire) that the weight-gradient kernels write into directly, sized for the
7 x ~153 GB/s point-to-point xGMI links, launched on a side stream as soon as the bucket's last gradient is produced so
they overlap the rest of backward.
"""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int, epoch: int = 0, shuffle: bool = False) -> List[int]:
    """DistributedSampler-style index striping (pixcontrast_18/contrast/data/__init__.py:21-25): pad to a multiple
    of `world`, then rank r takes items r, r+world, ..."""
    idx = list(range(n_items))
    if shuffle:
        g = torch.Generator()
        g.manual_seed(epoch)
        idx = torch.randperm(n_items, generator=g).tolist()
    total = (n_items + world - 1) // world * world
    idx += idx[: total - len(idx)]
    return idx[rank:total:world]


def grad_dest(p: torch.Tensor, shape=None, dtype=torch.float32):
    """The buffer a weight-gradient kernel of stswincl_amd.ops / headops should write p's gradient into, or None.

    A GradBucketReducer gives every parameter a persistent slice of its flat all-reduce bucket (`p._stswin_grad_dest`); a
    backward function that produces the WHOLE gradient of p with an overwriting kernel (gemm_tn(overwrite=True)) asks for it
    here and returns the view as the gradient, so autograd's AccumulateGrad adopts the view as p.grad and the bucket is
    already flat when its last gradient lands (no torch.cat / copy pass over the 500 MB of gradients per step).  None when
    there is no reducer, when p already holds a gradient (accumulation: the kernel would overwrite it), when the slice was
    already handed out in this backward (a parameter used twice), or on a shape / dtype mismatch - the caller then
    allocates as before and the reducer copies at launch time."""
    ent = getattr(p, "_stswin_grad_dest", None)
    if ent is None or p.grad is not None:
        return None
    red, view = ent[0](), ent[1]                        # (the reducer is held weakly: a dropped reducer must not live on here)
    if red is None:
        del p._stswin_grad_dest
        return None
    if red._accumulate_only or id(p) in red._claimed or view.dtype != dtype:
        return None
    if shape is not None and tuple(shape) != tuple(view.shape):
        if view.numel() != torch.Size(shape).numel():
            return None
        out = view.view(shape)
    else:
        out = view.view_as(view)                        # a FRESH tensor object: autograd only adopts a gradient nobody else holds
    red._claimed.add(id(p))
    return out


class _Done:
    def wait(self):
        return None


class GradBucketReducer:
    """Bucketed, overlapped all-reduce(mean) of parameter gradients.

    Parameters are bucketed in REVERSE registration order (the order backward produces them) into flat buffers that are
    allocated ONCE; every parameter owns a slice (`grad_dest`) that the weight-gradient kernels write directly.  Each
    parameter gets a post-accumulate-grad hook; when a bucket's last gradient lands, the gradients that are not already
    in place are copied into their slices (one multi-tensor copy), and the flat buffer is all-reduced asynchronously on
    `comm_stream` (GPU) while backward continues; `finish()` waits; p.grad then is the averaged slice.

    bucket_mb: 25-32 MB is the sweet spot on xGMI (7 point-to-point links of ~153 GB/s: a ring all-reduce of 32 MB over 8
    ranks moves 2 * 7/8 * 32 MB per rank = 56 MB, ~0.37 ms at link speed, against a launch latency of ~30 us: the first
    collective starts after ~1/16 of backward and ~16 are in flight per step of the 500 MB segmentation model).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 32.0, comm_dtype: Optional[torch.dtype] = None,
                 group=None, overlap: bool = True, simulate=None, hold_tn_fused: Optional[bool] = None):
        """simulate = (world, collective): measurement hook (tools/overlap_proxy.py) - behave like rank 0 of `world` ranks without a
        process group; collective(msg) is enqueued on the communication stream in place of dist.all_reduce (a stand-in kernel that
        holds compute units the way an RCCL kernel does) and returns None or an object with wait()."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self._simulate = simulate
        self.world = simulate[0] if simulate is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.comm_dtype = comm_dtype
        self.overlap = overlap
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, cur_bytes, cap = [], 0, bucket_mb * (1 << 20)
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= cap:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        # While an all-reduce runs on the comm stream its RCCL kernel holds compute units, so a 256-workgroup GEMM launched on the main
        # stream is not resident all at once.  The fused split-K combine of the weight-gradient GEMMs does not NEED residency (ticketed
        # row slices behind a bounded wait: include/stswin_hip.h), and with a stand-in collective holding 8 / 16 / 32 CUs the step
        # measures the same with it as with the separate pass (profiles/r05_overlap_proxy.txt) - but that is a single-GPU proxy, and no
        # N > 1 RCCL run of it exists.  Until one does, a real (non-simulated) overlapped reducer of a world > 1 takes the library's
        # ref-counted hold by default (separate combine pass, weight gradients launched one by one: the configuration every earlier
        # round's multi-rank tests ran); hold_tn_fused=False or STSWIN_DP_TN_FUSED=1 opts into the fused / grouped launches.
        if hold_tn_fused is None:
            hold_tn_fused = simulate is None and os.environ.get("STSWIN_DP_TN_FUSED", "0") != "1"
        self._tn_hold = None
        if self.world > 1 and overlap and hold_tn_fused:
            from . import hip
            self._tn_hold = hip.TnFusedHold()
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        # persistent flat buffers (parameter dtype) + per-parameter views; 64-byte aligned slices so that vector kernels can
        # write them.  With comm_dtype set the collective runs on a converted copy (gradient compression) instead.
        self._flat: List[torch.Tensor] = []
        self._views: List[List[torch.Tensor]] = []
        self._flat_numel: List[int] = []
        self._claimed = set()
        me = weakref.ref(self)
        for b in self.buckets:
            offs, tot = [], 0
            for p in b:
                offs.append(tot)
                tot += (p.numel() + 15) // 16 * 16
            dt = b[0].dtype
            assert all(p.dtype == dt for p in b), "GradBucketReducer: mixed parameter dtypes inside one bucket"
            self._flat_numel.append(tot)
            if self.world == 1:                   # nothing is reduced: no second copy of the gradients (~500 MB for the seg model)
                continue
            flat = torch.zeros(tot, dtype=dt, device=dev)
            self._flat.append(flat)
            vs = [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, b)]
            self._views.append(vs)
            for p, v in zip(b, vs):
                other = getattr(p, "_stswin_grad_dest", None)
                if other is not None and other[0]() is not None and other[0]() is not self:
                    raise RuntimeError("GradBucketReducer: this parameter already belongs to another live reducer - close() it first "
                                       "(two reducers would both hook the parameter and fight over its gradient slice)")
                p._stswin_grad_dest = (me, v)
        self._hooks = []
        self._accumulate_only = 0
        self.collectives = 0                      # all-reduce launches so far (tests count them)
        self.copied_bytes = 0                     # gradient bytes that had to be copied into their slices so far
        backend = dist.get_backend(group) if (dist.is_initialized() and simulate is None) else ""
        self._avg_op = dist.ReduceOp.AVG if backend == "nccl" else None
        if self.world > 1 and overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    def close(self):
        """Detach from the parameters (hooks and gradient destinations)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self._tn_hold is not None:
            self._tn_hold.release()
            self._tn_hold = None
        for p in self.params:
            ent = getattr(p, "_stswin_grad_dest", None)
            if ent is not None and ent[0]() in (self, None):
                del p._stswin_grad_dest

    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._comm = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._claimed = set()                     # (a backward that raises before finish() leaves its claims behind: call reset();
                                                  #  until then those parameters merely take the copy path, never a wrong slice)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward() calls inside this context only accumulate into p.grad (no bucket is counted
        or launched); the first backward() outside it reduces the accumulated gradients.  Without it a second backward()
        before finish() raises instead of silently dropping the later micro-batches."""
        self._accumulate_only += 1
        try:
            yield
        finally:
            self._accumulate_only -= 1

    def _on_grad(self, p):
        if self._accumulate_only:
            return
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] < 0:
            raise RuntimeError("GradBucketReducer: a parameter received a second gradient before finish() - one backward() "
                               "per finish(); wrap the earlier micro-batches of a gradient-accumulation step in no_sync()")
        if self._pending[i] == 0:
            self._launch(i)

    @staticmethod
    def _in_place(g: torch.Tensor, v: torch.Tensor) -> bool:
        return (g.dtype == v.dtype and g.shape == v.shape and g.is_contiguous() and g.data_ptr() == v.data_ptr()
                and g.device == v.device)

    def _launch(self, i):
        bucket = self.buckets[i]
        self._launched[i] = True
        if not bucket:
            return
        flat, views = self._flat[i], self._views[i]
        # Gradients the kernels already wrote into their slices need nothing.  The others (small torch-produced ones, layouts
        # the kernels cannot write) are copied in with one multi-tensor copy on the CALLING stream (it is ordered behind their
        # producers), zeros where this rank produced no gradient: the message is the same size on all ranks whatever their
        # grad-is-None pattern (a mismatch would hang or corrupt the collective).  p.grad then points at the slice.
        src, dst = [], []
        for p, v in zip(bucket, views):
            if p.grad is None:
                v.zero_()
            elif not self._in_place(p.grad, v):
                src.append(p.grad if p.grad.dtype == v.dtype else p.grad.to(v.dtype))
                dst.append(v)
                self.copied_bytes += v.numel() * v.element_size()
            if p.grad is None or not self._in_place(p.grad, v):
                p.grad = v.view_as(v)
        if src:
            torch._foreach_copy_(dst, src)
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            ctx = torch.cuda.stream(self.comm_stream)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            msg = flat if self.comm_dtype in (None, flat.dtype) else flat.to(self.comm_dtype)
            if self._simulate is not None:                 # measurement stand-in for the collective (no peers, values unchanged)
                self._comm[i] = msg
                self._work[i] = self._simulate[1](msg) or _Done()
                self.collectives += 1
                return
            if self._avg_op is not None:                   # RCCL: the division rides in the collective (one pass less)
                op = self._avg_op
            else:
                msg.div_(self.world)
                op = dist.ReduceOp.SUM
            self._comm[i] = msg
            self._work[i] = dist.all_reduce(msg, op=op, group=self.group, async_op=True)
            self.collectives += 1

    def finish(self):
        """Call after backward(): launches any bucket not yet reduced and makes the calling stream wait for all of them;
        every p.grad is its averaged slice of the bucket buffer afterwards (no copy back, except under comm_dtype
        compression, where the conversion back runs on the communication stream behind its collective)."""
        if self.world == 1:
            return
        for i in range(len(self.buckets)):
            if not self._launched[i]:
                self._launch(i)
        for i, work in enumerate(self._work):
            if work is None:
                continue
            if self.comm_stream is not None:
                with torch.cuda.stream(self.comm_stream):   # the collective was enqueued from comm_stream: wait there
                    work.wait()
                    if self._comm[i] is not self._flat[i]:
                        self._flat[i].copy_(self._comm[i])  # (still on comm_stream: ordered behind the all-reduce)
            else:
                work.wait()
                if self._comm[i] is not self._flat[i]:
                    self._flat[i].copy_(self._comm[i])
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()

    def bytes_per_step(self) -> int:
        """Bytes each rank contributes to the gradient all-reduce per step."""
        esz = torch.empty((), dtype=self.comm_dtype or torch.float32).element_size()
        return sum(self._flat_numel) * esz


def all_gather_embeddings(x: torch.Tensor, group=None) -> torch.Tensor:
    """Gather a (n, ...) tensor from every rank along dim 0 (the signature of the reference's unused
    ``dist_collect``, pixcontrast_18/contrast/util.py:47-58) - used by the optional inter-video key bank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return x
    out = [torch.empty_like(x) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, x.contiguous(), group=group)
    return torch.cat(out, dim=0)
