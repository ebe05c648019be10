"""Import-path aliases so the reference's training scripts find the MI355X-native modules under the dotted paths
they already use:

    net.Ours.swin_512 / base18 / ASPP / resnet,  utils.losses,  contrast.models.PixPro_swin_v5,
    contrast.models.Ours.{base,swin_tem,ASPPv5,resnet}

Call ``stswincl_amd.compat.install()`` before the script's own imports."""
from __future__ import annotations

import importlib
import sys
import types

_ALIASES = {
    "net.Ours.swin_512": "stswincl_amd.net.Ours.swin_512",
    "net.Ours.base18": "stswincl_amd.net.Ours.base18",
    "net.Ours.ASPP": "stswincl_amd.net.Ours.ASPP",
    "net.Ours.resnet": "stswincl_amd.net.Ours.resnet",
    "utils.losses": "stswincl_amd.utils.losses",
    "utils.LoadModel": "stswincl_amd.utils.LoadModel",
    "utils.EndoMetric": "stswincl_amd.utils.EndoMetric",
    "contrast.models.PixPro_swin_v5": "stswincl_amd.contrast.models.PixPro_swin_v5",
    "contrast.lars": "stswincl_amd.contrast.lars",
    "contrast.models.Ours.base": "stswincl_amd.contrast.models.Ours.base",
    "contrast.models.Ours.swin_tem": "stswincl_amd.net.Ours.swin_512",
    "contrast.models.Ours.ASPPv5": "stswincl_amd.net.Ours.ASPP",
    "contrast.models.Ours.resnet": "stswincl_amd.net.Ours.resnet",
}


def install(overwrite: bool = False) -> None:
    for alias, target in _ALIASES.items():
        parts = alias.split(".")
        for i in range(1, len(parts)):
            pkg = ".".join(parts[:i])
            if pkg not in sys.modules:
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
        if overwrite or alias not in sys.modules:
            mod = importlib.import_module(target)
            sys.modules[alias] = mod
            setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
