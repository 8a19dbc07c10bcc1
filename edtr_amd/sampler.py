"""Host-side mirror of reference utils/sampler.py: `space_timesteps` (:14-64) and `SpacedSampler` (:67-323) with the
same constructor, buffers, method names and argument meaning.  The per-step latent update is one libedtr_hip launch
(edtr_sampler_update) instead of ~10 elementwise ATen ops; the network evaluation is ControlLDM.forward."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch
from torch import nn

from . import ops
from .tiling import make_tiled_fn


def space_timesteps(num_timesteps: int, section_counts) -> set:
    """IDDPM respacing (guided-diffusion respace.py semantics, as the reference)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == desired:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += frac
        start += size
    return set(steps)


_TABLES = ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
           "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")


class SpacedSampler(nn.Module):
    def __init__(self, betas: np.ndarray) -> None:
        super().__init__()
        self.num_timesteps = len(betas)
        self.original_betas = betas
        self.original_alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        self.context = {}
        self._host_tables: Dict[str, np.ndarray] = {}
        self._coef_cache = None      # device copy of the per-step coefficient rows (edtr_sampler_update_indexed), per schedule

    def register(self, name: str, value: np.ndarray) -> None:
        self.register_buffer(name, torch.tensor(value, dtype=torch.float32))
        self._host_tables[name] = np.asarray(value, dtype=np.float64).astype(np.float32)

    def make_schedule(self, num_steps: int, used_timesteps=None) -> None:
        if used_timesteps is None:
            used_timesteps = space_timesteps(self.num_timesteps, str(num_steps))
        used = set(int(t) for t in used_timesteps)
        betas, last = [], 1.0
        for i, ac in enumerate(self.original_alphas_cumprod):
            if i in used:
                betas.append(1 - ac / last)
                last = ac
        assert len(betas) == num_steps
        self.timesteps = np.array(sorted(used), dtype=np.int32)
        betas = np.array(betas, dtype=np.float64)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        if num_steps == 1:
            post_logvar = np.array([-10.0])
        else:   # variance is 0 at the first step: clip the log with the second entry
            post_logvar = np.log(np.append(post_var[1], post_var[1:]))
        self.register("sqrt_recip_alphas_cumprod", np.sqrt(1.0 / ac))
        self.register("sqrt_recipm1_alphas_cumprod", np.sqrt(1.0 / ac - 1))
        self.register("posterior_variance", post_var)
        self.register("posterior_log_variance_clipped", post_logvar)
        self.register("posterior_mean_coef1", betas * np.sqrt(ac_prev) / (1.0 - ac))
        self.register("posterior_mean_coef2", (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac))
        self._coef_cache = None

    # -- algebra kept for API parity (reference :135-164); GPU tensors, torch indexing only -----------------
    def q_posterior_mean_variance(self, x_start, x_t, t):
        from .diffusion import extract_into_tensor
        mean = (extract_into_tensor(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + extract_into_tensor(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return (mean, extract_into_tensor(self.posterior_variance, t, x_t.shape),
                extract_into_tensor(self.posterior_log_variance_clipped, t, x_t.shape))

    def _predict_xstart_from_eps(self, x_t, t, eps):
        from .diffusion import extract_into_tensor
        return (extract_into_tensor(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - extract_into_tensor(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * eps)

    def predict_noise(self, model, x, t, cond, uncond, cfg_scale) -> torch.Tensor:
        if uncond is None or cfg_scale == 1.0:
            return model(x, t, cond)
        e_c = model(x, t, cond).contiguous()
        e_u = model(x, t, uncond).contiguous()
        out = torch.empty_like(e_c)     # uncond + s (cond - uncond)
        ops.launch(ops.make_axpby(x=e_c, y=e_u, a=cfg_scale, b=1.0 - cfg_scale, out=out, n=out.numel()))
        return out

    def _coefs(self, index: int) -> Tuple[float, float, float, float, float]:
        h = self._host_tables
        sigma = float(np.sqrt(h["posterior_variance"][index])) if index != 0 else 0.0
        return (float(h["sqrt_recip_alphas_cumprod"][index]), float(h["sqrt_recipm1_alphas_cumprod"][index]),
                float(h["posterior_mean_coef1"][index]), float(h["posterior_mean_coef2"][index]), sigma)

    @torch.no_grad()
    def p_sample(self, model, x, t, index, cond, uncond, cfg_scale):
        """eps -> x0 -> posterior mean -> x_{t-1} (reference :184-204).  `index` must be uniform over the batch (it
        always is: the loop builds it with torch.full_like)."""
        eps = self.predict_noise(model, x, t, cond, uncond, cfg_scale).contiguous().float()
        noise = torch.randn_like(x)     # drawn every step, masked on the last one (reference :199-203)
        x = x.contiguous().float()
        x_prev, pred_x0 = torch.empty_like(x), torch.empty_like(x)
        if torch.is_tensor(index) and index.device.type == "cuda":
            # a caller following the reference signature literally (index = torch.full_like(ts, ...), utils/sampler.py:311-312):
            # the coefficients are gathered on the device, per image, with no host round trip
            idx = index.reshape(-1).to(torch.int64)
            if idx.numel() == 1 and x.shape[0] > 1:
                idx = idx.expand(x.shape[0])
            if idx.numel() != x.shape[0]:       # the kernel reads index[image]: any other length would be an out-of-bounds device read
                raise ValueError(f"p_sample: index has {idx.numel()} entries for a batch of {x.shape[0]} (1 or the batch size)")
            ops.launch(ops.make_sampler_update_indexed(x=x, eps=eps, noise=noise.contiguous().float(), index=idx.contiguous(),
                                                       coefs=self._coef_table(x.device), x_prev=x_prev, pred_x0=pred_x0))
            return x_prev, pred_x0
        idx = int(index) if not torch.is_tensor(index) else int(index.reshape(-1)[0])      # host tensor / int: no device sync
        ops.launch(ops.make_sampler_update(x=x, eps=eps, noise=noise.contiguous().float(), coefs=self._coefs(idx),
                                           x_prev=x_prev, pred_x0=pred_x0, n=x.numel()))
        return x_prev, pred_x0

    def _coef_table(self, device) -> torch.Tensor:
        """[n_steps, 5] fp32 on the device: the rows _coefs() returns, for edtr_sampler_update_indexed (built per schedule)."""
        hit = self._coef_cache
        if hit is None or hit[0] != str(device):
            n = len(self._host_tables["posterior_variance"])
            hit = self._coef_cache = (str(device), torch.tensor([self._coefs(i) for i in range(n)], dtype=torch.float32).to(device))
        return hit[1]

    def _install_tiling(self, model, tile_size: int, tile_stride: int) -> None:
        # NB: like the reference (:288-303) the patched forward is never restored.
        forward = model.forward
        ctx_rep = {}      # (n) -> (source c_txt, its version, repeated copy): the SAME tensor object goes to every step, so the
        #                   engine's context cache (keyed on tensor identity) hits instead of re-projecting K / V^T per step

        def batched(x_tiles, windows, t, cond):
            # the windows of one latent stacked on the batch axis (window-major, like torch.cat of the per-window batches)
            n = len(windows)
            c_img = torch.cat([cond["c_img"][..., hi:he, wi:we] for hi, he, wi, we in windows], dim=0)
            c_txt = cond["c_txt"]
            hit = ctx_rep.get(n)
            if hit is None or hit[0] is not c_txt or hit[1] != c_txt._version:
                hit = ctx_rep[n] = (c_txt, c_txt._version, c_txt.repeat(n, 1, 1))
            return forward(x_tiles, t.repeat(n), {"c_txt": hit[2], "c_img": c_img})

        model.forward = make_tiled_fn(
            lambda x_tile, t, cond, hi, hi_end, wi, wi_end: forward(
                x_tile, t, {"c_txt": cond["c_txt"], "c_img": cond["c_img"][..., hi:hi_end, wi:wi_end]}),
            tile_size, tile_stride, batched_fn=batched)

    def _loop(self, model, device, img, batch_size, cond, uncond, cfg_scale, return_intermediates, progress=False, progress_leave=True):
        timesteps = np.flip(self.timesteps)
        total = len(self.timesteps)
        intermediates = []
        if progress:             # the reference's bar (utils/sampler.py:232,308: tqdm(..., leave=progress_leave, disable=not progress))
            try:
                from tqdm import tqdm
                timesteps = tqdm(timesteps, total=total, leave=progress_leave, desc="Spaced Sampler")
            except ImportError:  # tqdm is the reference's dependency, not this package's: without it the loop runs silently
                pass
        for i, step in enumerate(timesteps):
            ts = torch.full((batch_size,), int(step), device=device, dtype=torch.long)
            img, pred_x0 = self.p_sample(model, img, ts, total - i - 1, cond, uncond, cfg_scale)
            if return_intermediates:
                intermediates.append(pred_x0)
        return (img, intermediates) if return_intermediates else img

    @torch.no_grad()
    def sample(self, model, device, steps, batch_size, x_size, cond, uncond, cfg_scale, tiled=False, tile_size=-1,
               tile_stride=-1, x_T=None, progress=True, progress_leave=True, return_intermediates=False):
        self.make_schedule(steps)
        self.to(device)
        if tiled:
            self._install_tiling(model, tile_size, tile_stride)
        img = torch.randn((batch_size, *x_size), device=device) if x_T is None else x_T
        return self._loop(model, device, img, batch_size, cond, uncond, cfg_scale, return_intermediates, progress, progress_leave)

    @torch.no_grad()
    def manual_sample_with_timesteps(self, model, device, x_T, steps, used_timesteps, batch_size, cond, uncond, cfg_scale,
                                     tiled=False, tile_size=-1, tile_stride=-1, progress=True, progress_leave=True,
                                     return_intermediates=False):
        self.make_schedule(steps, used_timesteps)
        self.to(device)
        if tiled:
            self._install_tiling(model, tile_size, tile_stride)
        return self._loop(model, device, x_T, batch_size, cond, uncond, cfg_scale, return_intermediates, progress, progress_leave)
