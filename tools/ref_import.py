"""Make the read-only reference tree at /root/reference importable on CPU in THIS container.

Only used by tools/make_goldens.py (fixture generation) — never by tests, bench or the
product.  The reference needs four packages the image lacks (torchvision, omegaconf, ftfy,
timm); none of them is touched by the restoration hot path, so inert
stand-in modules are registered before the import.
"""
from __future__ import annotations

import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _module(name: str, **attrs) -> types.ModuleType:
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


class _PassThrough:
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, x, *args, **kwargs):
        return x


class _NoDropPath(torch.nn.Identity):
    def __init__(self, *args, **kwargs):
        super().__init__()


def install_stubs() -> None:
    if "torchvision" not in sys.modules:
        tv = _module("torchvision")
        tvt = _module("torchvision.transforms")
        tvt.transforms = _module("torchvision.transforms.transforms", Normalize=_PassThrough)
        tvt.functional = _module("torchvision.transforms.functional", normalize=lambda x, *a, **k: x)
        tv.transforms = tvt
        tv.models = _module("torchvision.models", get_model=lambda *a, **k: None)
    if "ftfy" not in sys.modules:
        _module("ftfy", fix_text=lambda s: s)
    if "timm" not in sys.modules:
        _module("timm")
        _module("timm.models")
        _module(
            "timm.models.layers",
            DropPath=_NoDropPath,
            trunc_normal_=torch.nn.init.trunc_normal_,
            to_2tuple=lambda v: v if isinstance(v, tuple) else (v, v),
        )
    if "omegaconf" not in sys.modules:
        _module("omegaconf")
        _module("omegaconf.listconfig", ListConfig=type("ListConfig", (list,), {}))


def import_reference():
    """Returns (ControlLDM, Diffusion, SpacedSampler, ref_common_module)."""
    sys.dont_write_bytecode = True
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import model  # noqa: F401  (reference package)
    from model.cldm import ControlLDM
    from model.gaussian_diffusion import Diffusion
    from utils.sampler import SpacedSampler
    import utils.common as ref_common
    return ControlLDM, Diffusion, SpacedSampler, ref_common
