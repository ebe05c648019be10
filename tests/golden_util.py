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
