"""Helpers shared by tests / smoke / bench: build a ControlLDM with the synthetic weights of edtr_amd.synth.
Nothing here touches the CPU oracle (oracle/ is imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline only)."""
from __future__ import annotations

import contextlib
from typing import Dict, Iterable

import torch

from . import arch, synth
from .model import ControlLDM


_SD_CACHE: Dict[str, Dict[str, Dict[str, torch.Tensor]]] = {}


def synthetic_state_dicts(cfg: dict, device=None, weights: str = "smooth") -> Dict[str, Dict[str, torch.Tensor]]:
    """{'unet': sd, 'controlnet': sd, 'vae': sd} with reference key names; hash keys carry the part prefix, exactly
    as tools/make_goldens.py did on the reference model (`cldm.state_dict()` keys).  Cached per configuration (hashing the
    1.3 G parameters of the SD-2.1 size takes about a minute of CPU; consumers copy the values, never mutate them)."""
    import json
    key = json.dumps({k: cfg[k] for k in ("unet_cfg", "controlnet_cfg", "vae_cfg")}, sort_keys=True, default=str) + str(device) + weights
    if key in _SD_CACHE:
        return _SD_CACHE[key]
    _SD_CACHE[key] = _synthetic_state_dicts(cfg, device, weights)
    return _SD_CACHE[key]


def _synthetic_state_dicts(cfg: dict, device=None, weights: str = "smooth") -> Dict[str, Dict[str, torch.Tensor]]:
    gen = synth.WEIGHT_SETS[weights]
    specs = {
        "unet": arch.unet_param_spec(arch.unet_arch(cfg["unet_cfg"])),
        "controlnet": arch.unet_param_spec(arch.unet_arch(cfg["controlnet_cfg"], controlnet=True)),
        "vae": arch.vae_param_spec(cfg["vae_cfg"]),
    }
    return {part: {k: gen(f"{part}.{k}", shp, device=device) for k, shp in spec} for part, spec in specs.items()}


def build_synthetic_cldm(cfg: dict, device, dtype=None, sds=None, precision=None, weights: str = "smooth") -> ControlLDM:
    from .model.params import skip_init
    on_gpu = torch.device(device).type == "cuda"
    with skip_init(), torch.device(device if on_gpu else "cpu"):      # parameters are born on the device: no 5 GB host round trip
        model = ControlLDM(**cfg)
    sds = sds or synthetic_state_dicts(cfg, device if on_gpu else None, weights)   # hashed on the device: bit-identical to the host
    model.unet.load_state_dict(sds["unet"], strict=True)
    model.load_controlnet_from_ckpt(sds["controlnet"])
    model.vae.load_state_dict(sds["vae"], strict=True)
    if dtype is not None:
        model.compute_dtype = dtype
    if precision is not None:
        model.precision = model.controlnet.precision = model.unet.precision = precision
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


def err_stats(a, b) -> Dict[str, float]:
    """Error of ``a`` against the reference ``b`` in the three norms the parity gates quote (BASELINE.md §3 asks for the "max
    relative error"; the relative L2 norm alone would hide a localised defect — one wrong halo column, one bad tile seam):
      l2     ||a - b||_2 / ||b||_2
      max    max|a - b| / max|b|                      (the worst element, relative to the signal's peak)
      p9999  99.99th percentile of |a - b| / max|b|   (the worst element outside 1e-4 of the tensor)"""
    a = torch.as_tensor(a).double().cpu().reshape(-1)
    b = torch.as_tensor(b).double().cpu().reshape(-1)
    d = (a - b).abs()
    peak = float(b.abs().max().clamp_min(1e-30))
    k = max(1, int(round(d.numel() * 1e-4)))
    p9999 = float(torch.topk(d, k).values[-1]) if d.numel() > 1 else float(d.max())
    return {"l2": float(d.norm() / b.norm().clamp_min(1e-30)), "max": float(d.max()) / peak, "p9999": p9999 / peak}
