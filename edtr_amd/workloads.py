"""The BASELINE.json configurations as functions of the reference-shaped API, shared by bench.py (timing) and the
full-size GPU parity tests (tests/test_gpu_fullsize.py) so that what is timed is what is checked:

  det512        configs[1]/[2]: vae_encode -> q_sample(t=200) -> 4 x (ControlNet + UNet + sampler update) -> vae_decode
                on a batch of 512x512 images (main/det/test_edtr.py:121-135)
  seg1024tiled  configs[3]: one 1024x1024 image, tiled VAE encoder (256-px tiles), latent-tiled denoiser (64/32 latent
                windows), untiled decoder (demo.py:96-124 with --vae-encoder-tiled --cldm-tiled)
  det512s50     configs[4] per GPU: 50-step spaced sampler from pure noise (utils/sampler.py:206-265), batch 4

Inputs are synthetic, generated for the GLOBAL batch with the closed-form hashes of edtr_amd.synth under `bench:*` names
and sliced per rank: tools/make_goldens.py feeds the same tensors to the reference to produce tests/golden/full_*.npz.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import synth
from .testing import injected_noise

USED_TIMESTEPS = [50, 100, 150, 200]
START_TIMESTEP = 200

WORKLOADS = {
    # name: (images per GPU, image size, denoise steps)
    "det512": (8, 512, 4),
    "seg1024tiled": (1, 1024, 4),
    "det512s50": (4, 512, 50),
}


@dataclass
class Inputs:
    pre_res: torch.Tensor                  # (B, 3, S, S) in [0, 1], this rank's slice
    c_txt: torch.Tensor                    # (B, 77, ctx_dim)
    noises: List[torch.Tensor]             # q_sample noise + one per denoise step (det512 / seg1024tiled); x_T for det512s50
    step_noises: List[torch.Tensor] = field(default_factory=list)   # det512s50 parity runs: 50 injected per-step noises
    t_start: Optional[torch.Tensor] = None


def make_inputs(workload: str, ctx_dim: int, dev, batch: int, size: int, rank: int = 0, world: int = 1,
                with_step_noises: bool = False) -> Inputs:
    from .parallel import shard_slice
    GB = batch * world
    sl = shard_slice(rank, world, GB)
    h = size // 8
    pre = synth.synth_input("bench:pre_res", (GB, 3, size, size), 0.0, 1.0)[sl].to(dev)
    c_txt = synth.synth_normal("bench:c_txt", (1, 77, ctx_dim)).expand(batch, -1, -1).contiguous().to(dev)
    noises = [synth.synth_normal(f"bench:noise{i}", (GB, 4, h, h))[sl].to(dev) for i in range(5)]
    step = []
    if workload == "det512s50" and with_step_noises:
        step = [synth.synth_normal(f"bench:s50noise{i}", (GB, 4, h, h))[sl].to(dev) for i in range(50)]
    return Inputs(pre, c_txt, noises, step, torch.full((batch,), START_TIMESTEP, dtype=torch.int64))   # host t: no device sync


def restore_pass(cldm, diffusion, sampler, inp: Inputs, workload: str, untiled_forward=None,
                 inject: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Dict[str, torch.Tensor]]:
    """One pass of the hot path over one batch.  Returns (decoded image in [-1, 1], final latent, {"z_pre": ...}).
    ``inject``: the sampler's per-step noise comes from ``inp.noises`` (parity runs: the tensors the reference golden was made with);
    False = `torch.randn_like` draws it on the GPU inside the pass, as the reference does (utils/sampler.py:199) — what bench.py times."""
    import contextlib
    dev = inp.pre_res.device
    B = inp.pre_res.shape[0]
    h, w = inp.pre_res.shape[2] // 8, inp.pre_res.shape[3] // 8
    tiled = workload == "seg1024tiled"
    if tiled:
        if untiled_forward is not None:     # the reference never restores the patched forward (sampler.py:288-303): re-arm it
            cldm.forward = untiled_forward
        z_pre = cldm.vae_encode(inp.pre_res * 2 - 1, sample=False, tiled=True, tile_size=256)
    else:
        z_pre = cldm.vae_encode(inp.pre_res * 2 - 1, sample=False)
    cond = {"c_txt": inp.c_txt, "c_img": z_pre}
    if workload == "det512s50":
        # DiffBIR-style: 50 spaced steps from pure noise; fresh torch.randn_like on the GPU per step unless a parity run injects it
        def run():
            return sampler.sample(model=cldm, device=dev, steps=50, batch_size=B, x_size=(4, h, w), cond=cond, uncond=None,
                                  cfg_scale=1.0, x_T=inp.noises[0], progress=False)
        if inp.step_noises and inject:
            with injected_noise(inp.step_noises):
                z = run()
        else:
            z = run()
        return cldm.vae_decode(z), z, {"z_pre": z_pre}
    x_T = diffusion.q_sample(z_pre, inp.t_start, inp.noises[0])
    with (injected_noise(inp.noises[1:]) if inject else contextlib.nullcontext()):
        z = sampler.manual_sample_with_timesteps(
            model=cldm, device=dev, x_T=x_T, steps=4, used_timesteps=USED_TIMESTEPS, batch_size=B, cond=cond, uncond=None,
            cfg_scale=1.0, progress=False, tiled=tiled, tile_size=64, tile_stride=32)
    return cldm.vae_decode(z), z, {"z_pre": z_pre}
