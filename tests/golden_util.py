"""Helpers shared by tools/gen_golden.py (build container) and the parity tests (everywhere).

Golden fixtures for full-width modules cannot carry their weights (ASPP alone is 16.7 M
parameters), so both sides regenerate them with :func:`det_fill`: a deterministic,
key-seeded fill that needs nothing from the reference.  The fixture then stores only
inputs (or their seed), the key/shape list and the reference's outputs.
"""
from __future__ import annotations

import os
import zlib
from typing import Dict

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gen(key: str, salt: int = 0) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


def det_tensor(key: str, shape, kind: str = "normal", scale: float = 1.0, salt: int = 0) -> torch.Tensor:
    g = _gen(key, salt)
    if kind == "normal":
        return torch.randn(*shape, generator=g) * scale
    if kind == "uniform":
        return torch.rand(*shape, generator=g) * scale
    raise ValueError(kind)


def det_fill(sd: Dict[str, torch.Tensor], salt: int = 0) -> Dict[str, torch.Tensor]:
    """Overwrite every floating tensor of a state-dict in place with key-seeded values.

    * integer buffers (relative_position_index, num_batches_tracked) and ``attn_mask`` keep
      their constructor values;
    * 1-D ``weight`` (BN/LN scale) ~ 1 + 0.1 N(0,1); ``running_var`` ~ U(0.5,1.5);
      ``running_mean`` and biases ~ 0.1 N(0,1); ``relative_position_bias_table`` ~ 0.5 N(0,1)
      (large enough that a wrong bias index is visible); everything else ~ N(0,1)/sqrt(fan_in).
    """
    for key, val in sd.items():
        if not torch.is_floating_point(val) or key.endswith("attn_mask"):
            continue
        shape = tuple(val.shape)
        if key.endswith("running_var"):
            new = 0.5 + det_tensor(key, shape, "uniform", 1.0, salt)
        elif key.endswith("running_mean") or key.endswith("bias"):
            new = det_tensor(key, shape, "normal", 0.1, salt)
        elif key.endswith("relative_position_bias_table"):
            new = det_tensor(key, shape, "normal", 0.5, salt)
        elif val.dim() == 1:
            new = 1.0 + det_tensor(key, shape, "normal", 0.1, salt)
        else:
            fan_in = int(np.prod(shape[1:]))
            new = det_tensor(key, shape, "normal", 1.0 / max(fan_in, 1) ** 0.5, salt)
        val.copy_(new.to(val.dtype))
    return sd


def load(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def to_sd(npz, prefix: str = "sd/") -> Dict[str, torch.Tensor]:
    """Pull the state-dict entries (stored as ``sd/<key>``) out of a fixture."""
    return {k[len(prefix):]: torch.from_numpy(np.array(npz[k])) for k in npz.files if k.startswith(prefix)}


def skeleton_sd(keys, shapes, dtypes) -> Dict[str, torch.Tensor]:
    """Rebuild an empty state-dict (key order preserved) from a fixture's key/shape/dtype lists."""
    sd = {}
    for k, s, d in zip(keys, shapes, dtypes):
        shape = tuple(int(v) for v in str(s).split("x") if v != "") if str(s) != "scalar" else ()
        sd[str(k)] = torch.zeros(shape, dtype=getattr(torch, str(d)))
    return sd


def shape_str(t: torch.Tensor) -> str:
    return "x".join(str(int(v)) for v in t.shape) if t.dim() else "scalar"


# ------------------------------------------------------------------ checkpoint-loader fixture
def toy_seg_model() -> torch.nn.Module:
    """A small module with TswinPlus's top-level attribute names (the loaders of seg18/utils/LoadModel.py only look at
    state-dict keys and shapes, so the fixture does not need the 60 M-parameter model).  Built identically by
    tools/gen_golden.py (which feeds it to the REFERENCE loaders) and by tests/test_loadmodel.py."""
    import torch.nn as nn

    class Swin(nn.Module):
        def __init__(self):
            super().__init__()
            self.qkv = nn.Linear(3, 6)
            self.register_buffer("attn_mask", torch.zeros(4, 2, 2))

    m = nn.Module()
    m.resnet = nn.Sequential(nn.Conv2d(2, 3, 1, bias=False), nn.BatchNorm2d(3))
    m.swin = Swin()
    m.aspp = nn.Sequential(nn.Conv2d(3, 2, 1))
    m.project1 = nn.Sequential(nn.Conv2d(3, 2, 1, bias=False), nn.BatchNorm2d(2))
    m.project2 = nn.Sequential(nn.Conv2d(3, 2, 1, bias=False), nn.BatchNorm2d(2))
    m.project3 = nn.Sequential(nn.Conv2d(6, 2, 1, bias=False), nn.BatchNorm2d(2))
    m.classifier = nn.Sequential(nn.Conv2d(6, 4, 3, padding=1, bias=False), nn.BatchNorm2d(4), nn.ReLU(), nn.Conv2d(4, 5, 1))
    m.module_list = nn.ModuleList([nn.Linear(2, 2)])
    det_fill(m.state_dict(), salt=11)          # (state_dict tensors alias the parameters)
    return m


def toy_checkpoints(model: torch.nn.Module):
    """-> {case: object to torch.save}: the on-disk shapes the reference writes (raw state-dict with / without the DataParallel
    prefix, seg18/utils/summary.py:76-88; the contrastive stage's {'opt', 'model', ...}, main_pretrain_swinv5.py:87-103),
    each with a shape mismatch (attn_mask of another resolution), a key the model lacks and a model key the file lacks."""
    import argparse
    sd = model.state_dict()
    val = lambda k, v: (det_tensor("ckpt/" + k, tuple(v.shape)) if v.is_floating_point() else v.clone() + 7)  # noqa: E731

    def raw(prefix):
        out = {}
        for k, v in sd.items():
            if k.startswith("classifier.3"):
                continue                                            # missing in the file: the model keeps its own
            kk = k if k.startswith("module_list") else prefix + k
            out[kk] = det_tensor("ckpt/" + k, (9, 2, 2)) if k.endswith("attn_mask") else val(k, v)
        out[prefix + "head.extra.weight"] = torch.ones(3)           # not in the model: dropped
        return out

    cl = {}
    names = (("pixpro.encoder_1", "resnet"), ("pixpro.encoder_2", "swin"), ("pixpro.encoder_3", "aspp"),
             ("pixpro.proj1", "project1"), ("pixpro.proj2", "project2"), ("pixpro.proj3", "project3"))
    for k, v in sd.items():
        for pre, dst in names:
            if k.startswith(dst + "."):
                cl[pre + k[len(dst):]] = det_tensor("ckpt/" + k, (9, 2, 2)) if k.endswith("attn_mask") else val(k, v)
    cl["pixpro.encoder_1_k.0.weight"] = torch.ones(3, 2, 1, 1)     # momentum branch: startswith('pixpro.encoder_1') too!
    cl["pixpro.projector.linear1.weight"] = torch.ones(3)
    cl["module.pixpro.proj1.0.weight"] = torch.ones(2, 3, 1, 1)     # a DDP-prefixed key matches no branch of the reference
    return {"raw_dataparallel": raw("module."), "raw_plain": raw(""),
            "contrastive": {"opt": argparse.Namespace(batch_size=8, amp_opt_level="O0"), "model": cl, "optimizer": {}, "scheduler": {},
                            "epoch": 3}}
