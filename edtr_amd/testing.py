"""Helpers shared by tests / smoke / bench: build a ControlLDM with the synthetic weights of edtr_amd.synth.
Nothing here touches the CPU oracle (oracle/ is imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline only)."""
from __future__ import annotations

import contextlib
from typing import Dict, Iterable

import torch

from . import arch, synth
from .model import ControlLDM


def synthetic_state_dicts(cfg: dict) -> Dict[str, Dict[str, torch.Tensor]]:
    """{'unet': sd, 'controlnet': sd, 'vae': sd} with reference key names; hash keys carry the part prefix, exactly
    as tools/make_goldens.py did on the reference model (`cldm.state_dict()` keys)."""
    specs = {
        "unet": arch.unet_param_spec(arch.unet_arch(cfg["unet_cfg"])),
        "controlnet": arch.unet_param_spec(arch.unet_arch(cfg["controlnet_cfg"], controlnet=True)),
        "vae": arch.vae_param_spec(cfg["vae_cfg"]),
    }
    return {part: {k: synth.synth_param(f"{part}.{k}", shp) for k, shp in spec} for part, spec in specs.items()}


def build_synthetic_cldm(cfg: dict, device, dtype=None, sds=None) -> ControlLDM:
    from .model.params import skip_init
    with skip_init():
        model = ControlLDM(**cfg)
    sds = sds or synthetic_state_dicts(cfg)
    model.unet.load_state_dict(sds["unet"], strict=True)
    model.load_controlnet_from_ckpt(sds["controlnet"])
    model.vae.load_state_dict(sds["vae"], strict=True)
    if dtype is not None:
        model.compute_dtype = dtype
    return model.eval().to(device)


@contextlib.contextmanager
def injected_noise(noises: Iterable[torch.Tensor]):
    """Replace torch.randn_like by a fixed list (the sampler draws one per step, reference utils/sampler.py:199)."""
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda x, *a, **k: next(it).to(device=x.device, dtype=x.dtype)
    try:
        yield
    finally:
        torch.randn_like = orig


def rel_err(a, b) -> float:
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
