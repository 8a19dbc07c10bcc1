"""Host-side mirror of reference model/gaussian_diffusion.py: `make_beta_schedule` (:9-31), `extract_into_tensor`
(:34-37) and the inference half of `Diffusion` (:40-84).  The training losses (:86-179) are out of scope."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
from torch import nn

from . import ops


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3) -> np.ndarray:
    if schedule == "linear":      # SD convention: linear in sqrt(beta)
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "cosine":
        ts = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        alphas = np.cos(ts / (1 + cosine_s) * np.pi / 2) ** 2
        alphas = alphas / alphas[0]
        return np.clip(1 - alphas[1:] / alphas[:-1], a_min=0, a_max=0.999)
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    raise ValueError(f"schedule '{schedule}' unknown.")


def extract_into_tensor(a: torch.Tensor, t: torch.Tensor, x_shape: Tuple[int, ...]) -> torch.Tensor:
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


class Diffusion(nn.Module):
    def __init__(self, timesteps=1000, beta_schedule="linear", loss_type="l2", linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, parameterization="eps"):
        super().__init__()
        assert parameterization in ["eps", "x0", "v"], "currently only supporting 'eps' and 'x0' and 'v'"
        self.num_timesteps, self.beta_schedule = timesteps, beta_schedule
        self.linear_start, self.linear_end, self.cosine_s = linear_start, linear_end, cosine_s
        self.parameterization, self.loss_type = parameterization, loss_type
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end,
                                   cosine_s=cosine_s)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.betas = betas
        self.register("sqrt_alphas_cumprod", np.sqrt(ac))
        self.register("sqrt_one_minus_alphas_cumprod", np.sqrt(1.0 - ac))
        self.register("sqrt_recip_alphas_cumprod", np.sqrt(1.0 / ac))
        self.register("sqrt_recipm1_alphas_cumprod", np.sqrt(1.0 / ac - 1))
        # host copies of the two q_sample tables: reading the device buffers would synchronise the stream
        self._host_sqrt_ac = np.sqrt(ac).astype(np.float32).tolist()
        self._host_sqrt_1mac = np.sqrt(1.0 - ac).astype(np.float32).tolist()

    def register(self, name: str, value: np.ndarray) -> None:
        self.register_buffer(name, torch.tensor(value, dtype=torch.float32))

    def q_sample(self, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """sqrt(ac_t) x0 + sqrt(1 - ac_t) noise, one libedtr_hip launch.  A GPU-resident ``t`` (every reference call
        site: demo.py:107-108, main/*/test_edtr.py) is read ON the device by edtr_q_sample — no ``.tolist()`` / host
        round trip; a host ``t`` is turned into scalars for edtr_axpby (per-sample launches when it is not uniform)."""
        if x_start.device.type != "cuda":
            raise RuntimeError("Diffusion.q_sample: GPU tensors required (no CPU fallback on the EDTR MI355X path)")
        x_start = x_start.contiguous().float()
        noise = noise.contiguous().float()
        out = torch.empty_like(x_start)
        if t.device.type == "cuda":
            if self.sqrt_alphas_cumprod.device != x_start.device:
                raise RuntimeError("Diffusion.q_sample: call diffusion.to(device) first (the schedule tables are on "
                                   f"{self.sqrt_alphas_cumprod.device})")
            ops.launch(ops.make_q_sample(x=x_start, noise=noise, t=t.to(torch.int64).contiguous(),
                                         tab_a=self.sqrt_alphas_cumprod, tab_b=self.sqrt_one_minus_alphas_cumprod, out=out))
            return out
        tl = t.tolist()
        a_tab, b_tab = self._host_sqrt_ac, self._host_sqrt_1mac
        if len(set(tl)) == 1:
            ops.launch(ops.make_axpby(x=x_start, y=noise, a=a_tab[tl[0]], b=b_tab[tl[0]], out=out, n=x_start.numel()))
        else:
            for i, ti in enumerate(tl):
                ops.launch(ops.make_axpby(x=x_start[i], y=noise[i], a=a_tab[ti], b=b_tab[ti], out=out[i],
                                          n=x_start[i].numel()))
        return out
