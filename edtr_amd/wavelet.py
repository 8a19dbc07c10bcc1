"""Wavelet colour fix applied to every restored image right after vae_decode (reference utils/common.py:99-147;
callers demo.py:124, main/*/test_edtr.py:135): content high-frequency + style low-frequency of a 5-level dilated
binomial decomposition.  10 libedtr_hip launches per call (edtr_wavelet_level) + one axpby."""
from __future__ import annotations

import torch

from . import ops


def wavelet_decomposition(image: torch.Tensor, levels: int = 5):
    """Returns (high_freq, low_freq) like the reference; fp32 NCHW GPU tensor in, new tensors out."""
    if image.device.type != "cuda":
        raise RuntimeError("wavelet_decomposition: GPU tensors required (no CPU fallback on the EDTR MI355X path)")
    x = image.contiguous().float()
    b, c, h, w = x.shape
    high = torch.zeros_like(x)
    cur, nxt = x, torch.empty_like(x)
    spare = torch.empty_like(x)
    for i in range(levels):
        ops.launch(ops.make_wavelet_level(src=cur, low=nxt, high=high, planes=b * c, H=h, W=w, radius=2 ** i))
        cur, nxt = nxt, (spare if cur is x else cur)
    return high, cur


def wavelet_reconstruction(content_feat: torch.Tensor, style_feat: torch.Tensor) -> torch.Tensor:
    high, _ = wavelet_decomposition(content_feat)
    _, low = wavelet_decomposition(style_feat)
    out = torch.empty_like(high)
    ops.launch(ops.make_axpby(x=high, y=low, a=1.0, b=1.0, out=out, n=out.numel()))
    return out
