"""Import-path shim: make the reference's own dotted names resolve to the MI355X implementation.

The reference builds its models from YAML `target:` strings through `utils.common.instantiate_from_config`
(reference utils/common.py:23-34; e.g. `target: model.cldm.ControlLDM`, configs/det/demo.yaml:21) and its scripts
import `from model import ControlLDM, Diffusion` / `from utils.sampler import SpacedSampler` (demo.py:14-21,
main/det/test_edtr.py:10-21).  Two ways to switch those names over without editing a script:

* standalone: put `<repo>/shim` in front of `sys.path` — `shim/model` and `shim/utils` are real packages that re-export
  the edtr_amd classes at the reference's module paths;
* overlay on a reference checkout: `edtr_amd.shim.install()` pre-seeds `sys.modules` with the hot-path modules
  (`model.cldm`, `model.controlnet`, `model.vae`, `model.gaussian_diffusion`, `utils.sampler`) and patches the matching
  attributes of the already-importable reference packages, so `model.resnet`, `utils.detection`, ... keep coming from
  the reference while the restoration path runs on libedtr_hip.
"""
from __future__ import annotations

import importlib
import sys
import types
from typing import Any, Mapping

HOT_PATH_MODULES = ("model.cldm", "model.controlnet", "model.vae", "model.gaussian_diffusion", "model.clip", "model.swinir",
                    "utils.sampler")


def get_obj_from_str(string: str, reload: bool = False) -> Any:
    """reference utils/common.py:23-28."""
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config: Mapping[str, Any]) -> Any:
    """reference utils/common.py:31-34 (same KeyError when `target` is missing)."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def _exports() -> dict:
    from . import diffusion, sampler
    from .model import cldm, clip, swinir
    return {
        "model.cldm": dict(ControlLDM=cldm.ControlLDM, disabled_train=cldm.disabled_train),
        "model.controlnet": dict(ControlledUnetModel=cldm.ControlledUnetModel, ControlNet=cldm.ControlNet),
        "model.vae": dict(AutoencoderKL=cldm.AutoencoderKL),
        "model.clip": dict(FrozenOpenCLIPEmbedder=clip.FrozenOpenCLIPEmbedder),
        "model.swinir": dict(SwinIR=swinir.SwinIR),
        "model.gaussian_diffusion": dict(Diffusion=diffusion.Diffusion, make_beta_schedule=diffusion.make_beta_schedule,
                                         extract_into_tensor=diffusion.extract_into_tensor),
        "utils.sampler": dict(SpacedSampler=sampler.SpacedSampler, space_timesteps=sampler.space_timesteps),
    }


def install(overlay: bool = True) -> None:
    """Register the hot-path modules under the reference's dotted names.  With ``overlay`` the parent packages
    (`model`, `utils`) are imported first when they exist (a reference checkout on sys.path) and their re-exported
    class attributes (`model.ControlLDM`, `model.Diffusion`, ...) are repointed; without a reference checkout empty
    parent packages are created."""
    exports = _exports()
    for parent in ("model", "utils"):
        if parent not in sys.modules:
            try:
                if not overlay:
                    raise ImportError
                importlib.import_module(parent)
            except Exception:
                pkg = types.ModuleType(parent)
                pkg.__path__ = []          # a package, so that `import model.cldm` consults sys.modules
                sys.modules[parent] = pkg
    for name, attrs in exports.items():
        mod = types.ModuleType(name)
        mod.__dict__.update(attrs)
        mod.__edtr_amd_shim__ = True
        sys.modules[name] = mod
        parent, leaf = name.split(".")
        setattr(sys.modules[parent], leaf, mod)
        for k, v in attrs.items():
            if hasattr(sys.modules[parent], k) or parent == "model":
                setattr(sys.modules[parent], k, v)     # `from model import ControlLDM` (model/__init__.py re-exports)
