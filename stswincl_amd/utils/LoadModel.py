"""Checkpoint compatibility: the reference's on-disk formats either side of the path.

* seg stages save a raw ``state_dict`` (``*.t7``), possibly ``module.``-prefixed by nn.DataParallel
  (seg18/utils/summary.py:76-88, train_swin.py:268-272);
* the contrastive stage saves ``{'model': state_dict, 'optimizer': ..., 'epoch': ...}`` with ``pixpro.encoder_1/2/3`` and
  ``pixpro.proj1/2/3`` prefixes (pixcontrast_18/main_pretrain_swinv5.py:87-103);
* ``load_model_mswin_CL`` (seg18/utils/LoadModel.py:6-49) maps the latter onto ``TswinPlus`` and silently keeps the
  model's own tensor wherever shapes differ (e.g. ``attn_mask`` when the resolution changed).

Same function names and key handling as seg18/utils/LoadModel.py: ``load_model`` strips ``module.`` only from
``module.resnet*`` keys (:55-59), ``load_model_full`` strips nothing (:96-100), ``load_model_full_fortest`` strips it from
every ``module*`` key (:128-132), ``load_model_mswin_CL`` remaps the contrastive prefixes (:14-28); all four keep the model's
own tensor where shapes differ; which keys each loader takes from which file layout is pinned against the reference's own
functions (tests/golden/loadmodel.npz, tools/gen_golden.py --only loadmodel).  Deviations: ``map_location`` is the model's
device instead of a hard-coded 'cuda:0'; the GEMM operand caches are dropped after a load (ops.clear_caches).

The reference's contrastive checkpoints hold ``{'opt': argparse.Namespace, 'model', 'optimizer', 'scheduler', 'epoch'}``
(main_pretrain_swinv5.py:91-102); torch >= 2.6 refuses the Namespace under its default ``weights_only=True``, so
``_torch_load`` allow-lists it; any other global is refused unless the caller opts in (``trusted=True`` /
``STSWIN_TRUST_CHECKPOINTS=1``).
"""
from __future__ import annotations

from collections import OrderedDict

import torch

_CL_PREFIXES = (("pixpro.encoder_1", "resnet"), ("pixpro.encoder_2", "swin"), ("pixpro.encoder_3", "aspp"),
                ("pixpro.proj1", "project1"), ("pixpro.proj2", "project2"), ("pixpro.proj3", "project3"))


def _torch_load(path, map_location, trusted=None):
    """torch.load with the safe unpickler (``weights_only=True``; ``argparse.Namespace`` - the ``'opt'`` entry of the
    reference's contrastive checkpoints - is the only extra global allowed).  A checkpoint that needs other globals is
    refused with the unpickler's message unless the caller opts in: ``trusted=True`` or ``STSWIN_TRUST_CHECKPOINTS=1``
    (full pickle: runs whatever code the file holds - only for checkpoints you wrote yourself)."""
    import argparse
    import os
    import pickle
    if trusted is None:
        trusted = os.environ.get("STSWIN_TRUST_CHECKPOINTS") == "1"
    # Only a REFUSAL of the safe unpickler takes the fallback / gets the "refused" message: UnpicklingError (a global that is not on the
    # allow-list), or one of the errors it raises for legacy formats and classes that cannot be imported here - recognised by their
    # message.  Anything else (a truncated or corrupt file, a map_location / device error, an I/O error) is re-raised unchanged: retrying
    # it with the unrestricted unpickler would run the file's code for nothing, and calling it "refused" would send the user to
    # STSWIN_TRUST_CHECKPOINTS=1 for a problem that flag cannot fix (round-4 advisor).
    def is_refusal(e: BaseException) -> bool:
        if isinstance(e, pickle.UnpicklingError):
            return True
        msg = str(e)
        hints = ("weights_only", "Weights only", "Unsupported global", "unsupported global", "legacy", "torch.serialization.add_safe_globals",
                 "safe_globals", "was not an allowed global", "Can't get attribute", "No module named")
        return isinstance(e, (RuntimeError, AttributeError, ModuleNotFoundError)) and any(h in msg for h in hints)

    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError, ModuleNotFoundError) as e:
        if not is_refusal(e):
            raise
        if trusted:
            return torch.load(path, map_location=map_location, weights_only=False)
        raise pickle.UnpicklingError(
            f"{path}: refused by the safe unpickler ({type(e).__name__}: {e}).  If this checkpoint comes from a source you trust, pass "
            "trusted=True to the loader or set STSWIN_TRUST_CHECKPOINTS=1 to unpickle it without restrictions.") from e


def _device_of(model):
    try:
        return next(model.parameters()).device
    except StopIteration:
        return torch.device("cpu")


def strip_module_prefix(state_dict):
    """nn.DataParallel / DDP checkpoints carry a ``module.`` prefix."""
    return OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())


def remap_contrastive_keys(cl_state_dict):
    """``pixpro.encoder_1.* -> resnet.*`` etc. (seg18/utils/LoadModel.py:14-28); other keys are dropped."""
    out = OrderedDict()
    for key, val in cl_state_dict.items():          # (a DDP 'module.' prefix matches no branch there: the file is saved from model.module)
        for src, dst in _CL_PREFIXES:
            if key.startswith(src):
                out[dst + key[len(src):]] = val
                break
    return out


def _merge_into(model, state_dict, log=True):
    """Shape-checked merge (LoadModel.py:30-46): mismatching or missing keys keep the model's own tensors."""
    own = model.state_dict()
    merged = OrderedDict()
    for key, val in state_dict.items():
        if key in own:
            if tuple(val.shape) != tuple(own[key].shape):
                if log:
                    print('Skip loading parameter {}, required shape{}, loaded shape{}.'.format(key, own[key].shape, val.shape))
                merged[key] = own[key]
            else:
                merged[key] = val
        elif log:
            print('Drop parameter {}.'.format(key))
    for key in own:
        if key not in merged:
            if log:
                print('No param {}.'.format(key))
            merged[key] = own[key]
    model.load_state_dict(merged, strict=False)
    return model


def _after_load(model):
    from .. import ops
    ops.clear_caches()            # cached bf16 / transposed GEMM operands of the overwritten parameters
    return model


def _raw_state_dict(pretrain_dir, model, trusted=None):
    ckpt = _torch_load(pretrain_dir, _device_of(model), trusted)
    print('loaded pretrained weights form %s !' % pretrain_dir)
    return ckpt


def load_model_mswin_CL(model, pretrain_dir, log=True, trusted=None):
    ckpt = _torch_load(pretrain_dir, _device_of(model), trusted)
    print('loaded pretrained weights form %s !' % pretrain_dir)
    return _after_load(_merge_into(model, remap_contrastive_keys(ckpt['model']), log))


def load_model(model, pretrain_dir, log=True, trusted=None):
    """Raw seg checkpoint; only ``module.resnet*`` keys lose their DataParallel prefix (LoadModel.py:55-59)."""
    sd = _raw_state_dict(pretrain_dir, model, trusted)
    sd = OrderedDict((k[7:] if (k.startswith('module.resnet') and not k.startswith('module_list')) else k, v) for k, v in sd.items())
    return _after_load(_merge_into(model, sd, log))


def load_model_full(model, pretrain_dir, log=True, trusted=None):
    """Raw checkpoint, keys taken as they are (LoadModel.py:96-100)."""
    return _after_load(_merge_into(model, OrderedDict(_raw_state_dict(pretrain_dir, model, trusted)), log))


def load_model_full_fortest(model, pretrain_dir, log=True, trusted=None):
    """Raw checkpoint saved from nn.DataParallel: every ``module*`` key loses its first 7 characters (LoadModel.py:128-132)."""
    sd = _raw_state_dict(pretrain_dir, model, trusted)
    sd = OrderedDict((k[7:] if (k.startswith('module') and not k.startswith('module_list')) else k, v) for k, v in sd.items())
    return _after_load(_merge_into(model, sd, log))
