"""Checkpoint compatibility: the reference's on-disk formats either side of the path.

* seg stages save a raw ``state_dict`` (``*.t7``), possibly ``module.``-prefixed by nn.DataParallel
  (seg18/utils/summary.py:76-88, train_swin.py:268-272);
* the contrastive stage saves ``{'model': state_dict, 'optimizer': ..., 'epoch': ...}`` with ``pixpro.encoder_1/2/3`` and
  ``pixpro.proj1/2/3`` prefixes (pixcontrast_18/main_pretrain_swinv5.py:87-103);
* ``load_model_mswin_CL`` (seg18/utils/LoadModel.py:6-49) maps the latter onto ``TswinPlus`` and silently keeps the
  model's own tensor wherever shapes differ (e.g. ``attn_mask`` when the resolution changed).

Same function names and behaviour; ``map_location`` defaults to the model's device instead of a hard-coded 'cuda:0'.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

_CL_PREFIXES = (("pixpro.encoder_1", "resnet"), ("pixpro.encoder_2", "swin"), ("pixpro.encoder_3", "aspp"),
                ("pixpro.proj1", "project1"), ("pixpro.proj2", "project2"), ("pixpro.proj3", "project3"))


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
    for key, val in cl_state_dict.items():
        key = key[7:] if key.startswith("module.") else key
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


def load_model_mswin_CL(model, pretrain_dir, log=True):
    ckpt = torch.load(pretrain_dir, map_location=_device_of(model))
    print('loaded pretrained weights form %s !' % pretrain_dir)
    return _merge_into(model, remap_contrastive_keys(ckpt['model']), log)


def load_model(model, pretrain_dir, log=True):
    """Raw seg checkpoint (optionally ``module.``-prefixed), shape-checked."""
    ckpt = torch.load(pretrain_dir, map_location=_device_of(model))
    print('loaded pretrained weights form %s !' % pretrain_dir)
    if isinstance(ckpt, dict) and 'model' in ckpt and not any(k.endswith('.weight') for k in ckpt):
        ckpt = ckpt['model']
    return _merge_into(model, strip_module_prefix(ckpt), log)


load_model_full = load_model
