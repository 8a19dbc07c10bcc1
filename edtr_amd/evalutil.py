"""Evaluation-harness helpers around the restoration path: host-side mirrors of the reference's
batch padding (`utils/detection.py:141-165`) and PSNR metric (`utils/common.py:194-247`), plus an accelerate-free
data-parallel driver that shards a list of pre-restored images over the ranks of `torch.distributed`, runs the
edtr_amd path on each shard and reports PSNR.  Plain tensor bookkeeping: no kernels of their own."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from .parallel import shard_slice


def list_to_batch(img_list: Sequence[torch.Tensor], img_size: int, device) -> torch.Tensor:
    """Zero-pad every (C, H, W) image at the bottom / right to (C, img_size, img_size) and stack (detection.py:141-157)."""
    out = []
    for img in img_list:
        ph, pw = img_size - img.size(1), img_size - img.size(2)
        out.append(torch.nn.functional.pad(img.unsqueeze(0).to(device), pad=(0, pw, 0, ph), mode="constant"))
    return torch.cat(out, dim=0) if out else torch.Tensor().to(device)


def batch_to_list(img_batch: torch.Tensor, img_list: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """Crop every batch entry back to the size of the matching list entry (detection.py:160-165)."""
    return [img_batch[i][:, :img.size(1), :img.size(2)] for i, img in enumerate(img_list)]


def pad_if_smaller(imgs: torch.Tensor, size: int) -> torch.Tensor:
    """Zero-pad (n, c, h, w) at the bottom / right up to ``size`` in each dimension that is smaller (utils/common.py:337-340)."""
    _, _, h, w = imgs.size()
    return torch.nn.functional.pad(imgs, pad=(0, max(size - w, 0), 0, max(size - h, 0)), mode="constant", value=0)


def pad_to_multiples_of(imgs: torch.Tensor, multiple: int) -> torch.Tensor:
    """Zero-pad (n, c, h, w) at the bottom / right to the next multiples of ``multiple`` (utils/common.py:343-348)."""
    _, _, h, w = imgs.size()
    if h % multiple == 0 and w % multiple == 0:
        return imgs.clone()
    ph, pw = ((x + multiple - 1) // multiple * multiple - x for x in (h, w))
    return torch.nn.functional.pad(imgs, pad=(0, pw, 0, ph), mode="constant", value=0)


def rgb2ycbcr_pt(img: torch.Tensor, y_only: bool = False) -> torch.Tensor:
    """ITU-R BT.601 RGB -> YCbCr on (n, 3, h, w) in [0, 1] (common.py:194-216)."""
    if y_only:
        weight = torch.tensor([[65.481], [128.553], [24.966]]).to(img)
        out = torch.matmul(img.permute(0, 2, 3, 1), weight).permute(0, 3, 1, 2) + 16.0
    else:
        weight = torch.tensor([[65.481, -37.797, 112.0], [128.553, -74.203, -93.786], [24.966, 112.0, -18.214]]).to(img)
        bias = torch.tensor([16, 128, 128]).view(1, 3, 1, 1).to(img)
        out = torch.matmul(img.permute(0, 2, 3, 1), weight).permute(0, 3, 1, 2) + bias
    return out / 255.0


def calculate_psnr_pt(img: torch.Tensor, img2: torch.Tensor, crop_border: int, test_y_channel: bool = False) -> torch.Tensor:
    """Per-image PSNR in dB of (n, 3/1, h, w) tensors in [0, 1] (common.py:219-247)."""
    assert img.shape == img2.shape, f"Image shapes are different: {img.shape}, {img2.shape}."
    if crop_border != 0:
        img = img[:, :, crop_border:-crop_border, crop_border:-crop_border]
        img2 = img2[:, :, crop_border:-crop_border, crop_border:-crop_border]
    if test_y_channel:
        img, img2 = rgb2ycbcr_pt(img, y_only=True), rgb2ycbcr_pt(img2, y_only=True)
    img, img2 = img.to(torch.float64), img2.to(torch.float64)
    mse = torch.mean((img - img2) ** 2, dim=[1, 2, 3])
    return 10.0 * torch.log10(1.0 / (mse + 1e-8))


@torch.no_grad()
def restore_dataset(cldm, diffusion, sampler, pre_restored: Sequence[torch.Tensor], gts: Optional[Sequence[torch.Tensor]] = None,
                    img_size: int = 512, batch_size: int = 8, used_timesteps=(50, 100, 150, 200), start_timestep: int = 200,
                    colour_fix: bool = True, swinir=None, pad_mode: str = "batch", multiple: int = 64,
                    clamp: bool = True) -> Tuple[List[torch.Tensor], Optional[torch.Tensor]]:
    """The restoration loop of main/det/test_edtr.py:121-135 without accelerate: this rank's shard of the
    (C, h, w <= img_size) pre-restored images is padded, pushed through vae_encode -> q_sample(t) -> spaced sampler ->
    vae_decode (-> wavelet colour fix), cropped back, and — when ground truth is given — scored with PSNR; the scalar
    PSNR sums are all-reduced, nothing else crosses ranks.  With ``swinir`` (an edtr_amd.model.SwinIR) the inputs are the
    low-quality images themselves and the pre-restoration runs on the padded batch first (`cfg.model.pre_restoration`,
    main/det/test_edtr.py:118).  ``pad_mode="demo"`` is the single-image flow of demo.py:84-131,165 instead: every image on its own,
    `pad_if_smaller(img_size)` -> `pad_to_multiples_of(multiple)` -> (SwinIR) -> the same path -> crop back to the input's size
    (``clamp=False`` keeps the values the reference hands to `save_image`).  Returns (restored images of this shard, mean PSNR or None)."""
    import torch.distributed as dist
    from .wavelet import wavelet_reconstruction
    dev = next(cldm.unet.parameters()).device
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    sl = shard_slice(rank, world, len(pre_restored))
    mine = list(pre_restored[sl])
    outs: List[torch.Tensor] = []
    if pad_mode not in ("batch", "demo"):
        raise ValueError(f"pad_mode must be 'batch' or 'demo', got {pad_mode!r}")
    step = 1 if pad_mode == "demo" else batch_size
    for i in range(0, len(mine), step):
        chunk = mine[i:i + step]
        if pad_mode == "demo":
            pre = pad_to_multiples_of(pad_if_smaller(chunk[0][None].to(dev).float(), img_size), multiple)
        else:
            pre = list_to_batch(chunk, img_size, dev).float()
        if swinir is not None:
            pre = swinir(pre)
        cond = cldm.prepare_condition(pre, [""] * pre.size(0))
        t = torch.full((pre.size(0),), start_timestep, dtype=torch.int64)
        x_T = diffusion.q_sample(cond["c_img"], t, torch.randn_like(cond["c_img"]))
        z = sampler.manual_sample_with_timesteps(model=cldm, device=dev, x_T=x_T, steps=len(used_timesteps),
                                                 used_timesteps=list(used_timesteps), batch_size=pre.size(0), cond=cond,
                                                 uncond=None, cfg_scale=1.0, progress=False)
        res = (cldm.vae_decode(z) + 1) / 2
        if colour_fix:
            res = wavelet_reconstruction(res, pre)
        outs.extend(batch_to_list(res.clamp(0, 1) if clamp else res, chunk))
    psnr = None
    if gts is not None:
        mine_gt = list(gts[sl])
        acc = torch.zeros(2, dtype=torch.float64, device=dev)
        for o, g in zip(outs, mine_gt):
            acc[0] += calculate_psnr_pt(o[None].float(), g[None].to(dev).float(), crop_border=0)[0]
            acc[1] += 1
        if world > 1:
            dist.all_reduce(acc)
        psnr = acc[0] / acc[1].clamp_min(1)
    return outs, psnr
