"""Data-parallel gradient averaging for the seg / contrastive training steps: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference uses nn.DataParallel (seg18/train_swin.py:131-135) or DDP (pixcontrast_18/main_pretrain_swinv5.py:54);
both reduce to: every rank holds the full model, processes its shard of clips, and gradients are averaged.
Clips are independent, so the only collective on the data path is this all-reduce.  Buckets are flat fp32 (or
bf16) buffers sized for the 7 x ~153 GB/s point-to-point xGMI links (few, large messages), launched on a side
stream as soon as the bucket's last gradient is produced so they overlap the rest of backward.
"""
from __future__ import annotations

import contextlib
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


class GradBucketReducer:
    """Bucketed, overlapped all-reduce(mean) of parameter gradients.

    Parameters are bucketed in REVERSE registration order (the order backward produces them).  Each parameter gets a
    post-accumulate-grad hook; when a bucket's last gradient lands the bucket is flattened and all-reduced
    asynchronously on `comm_stream` (GPU) while backward continues; `finish()` waits and scatters the averages back.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, comm_dtype: Optional[torch.dtype] = None,
                 group=None, overlap: bool = True):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
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
        self._pending = [0] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._flat = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._hooks = []
        self._accumulate_only = 0
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self._avg_op = dist.ReduceOp.AVG if backend == "nccl" else None
        if self.world > 1 and overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    def reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._flat = [None] * len(self.buckets)
        self._launched = [False] * len(self.buckets)

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

    def _launch(self, i):
        bucket = self.buckets[i]
        self._launched[i] = True
        if not bucket:
            return
        dt = self.comm_dtype or torch.float32
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            ctx = torch.cuda.stream(self.comm_stream)
        else:
            ctx = contextlib.nullcontext()
        with ctx:
            # every parameter of the bucket takes part, zeros where this rank produced no gradient: the message size is
            # the same on all ranks whatever their grad-is-None pattern (a mismatch would hang or corrupt the collective)
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt) for p in bucket])
            if self._avg_op is not None:                   # RCCL: the division rides in the collective (one pass less)
                op = self._avg_op
            else:
                flat.div_(self.world)
                op = dist.ReduceOp.SUM
            self._flat[i] = (flat, bucket)
            self._work[i] = dist.all_reduce(flat, op=op, group=self.group, async_op=True)

    def finish(self):
        """Call after backward(): launches any bucket not yet reduced, waits, and points every p.grad at its averaged
        slice of the bucket buffer (a pointer swap: no copy back)."""
        if self.world == 1:
            return
        for i in range(len(self.buckets)):
            if not self._launched[i]:
                self._launch(i)
        for i, item in enumerate(self._flat):
            if item is None:
                continue
            flat, bucket = item
            if self.comm_stream is not None:
                with torch.cuda.stream(self.comm_stream):   # the collective was enqueued from comm_stream: wait there
                    self._work[i].wait()
            else:
                self._work[i].wait()
            self._unflatten(flat, bucket)
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()

    @staticmethod
    def _unflatten(flat, bucket):
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view(p.shape)
            p.grad = g if g.dtype == p.dtype else g.to(p.dtype)
            off += n

    def bytes_per_step(self) -> int:
        """Bytes each rank contributes to the gradient all-reduce per step."""
        esz = torch.empty((), dtype=self.comm_dtype or torch.float32).element_size()
        return sum(p.numel() for p in self.params) * esz


def all_gather_embeddings(x: torch.Tensor, group=None) -> torch.Tensor:
    """Gather a (n, ...) tensor from every rank along dim 0 (the signature of the reference's unused
    ``dist_collect``, pixcontrast_18/contrast/util.py:47-58) - used by the optional inter-video key bank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return x
    out = [torch.empty_like(x) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, x.contiguous(), group=group)
    return torch.cat(out, dim=0)
