"""Latent tiling of a model forward with Gaussian-weighted overlap-add: host-side mirror of reference
utils/common.py `sliding_windows` (:351-364), `gaussian_weights` (:151-165) and `make_tiled_fn` (:367-427), with the
accumulate / normalise steps as libedtr_hip launches (edtr_tile_accumulate, edtr_divide)."""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np
import torch

from . import ops


def sliding_windows(h: int, w: int, tile_size: int, tile_stride: int) -> List[Tuple[int, int, int, int]]:
    def starts(n: int) -> List[int]:
        s = list(range(0, n - tile_size + 1, tile_stride))
        if (n - tile_size) % tile_stride != 0:
            s.append(n - tile_size)     # last window snapped to the edge
        return s
    return [(hi, hi + tile_size, wi, wi + tile_size) for hi in starts(h) for wi in starts(w)]


def gaussian_weights(tile_width: int, tile_height: int) -> np.ndarray:
    """var = 0.01; the x midpoint is (w-1)/2 while the y midpoint is h/2 — kept as in the reference."""
    var = 0.01
    xs, ys = np.arange(tile_width, dtype=np.float64), np.arange(tile_height, dtype=np.float64)
    norm = np.sqrt(2 * np.pi * var)
    xp = np.exp(-(xs - (tile_width - 1) / 2) ** 2 / (tile_width * tile_width) / (2 * var)) / norm
    yp = np.exp(-(ys - tile_height / 2) ** 2 / (tile_height * tile_height) / (2 * var)) / norm
    return np.outer(yp, xp)


def make_tiled_fn(fn: Callable, size: int, stride: int, weight: str = "gaussian", batched_fn: Callable = None,
                  max_batch: int = 16) -> Callable:
    """Only the first argument (the latent, fp32 NCHW) is split; `fn` receives the tile plus hi/hi_end/wi/wi_end
    keyword arguments when it has extra arguments (reference utils/common.py:413-414).

    MI355X addition: all windows have the same size, and the wrapped network treats batch entries independently, so
    when ``batched_fn(x_tiles, windows, *args, **kwargs)`` is given the windows are evaluated in groups stacked on the
    batch axis (<= max_batch entries per call) instead of one forward per window — the same numbers, several times the
    work per launch (9 windows of a 1024x1024 image become one batch-9 forward per denoise step)."""

    def tiled_fn(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        b, c, h, w = x.shape
        out = torch.zeros((b, c, h, w), dtype=torch.float32, device=x.device)
        count = torch.zeros_like(out)
        wts_np = gaussian_weights(size, size) if weight == "gaussian" else np.ones((size, size))
        wts = torch.tensor(wts_np, dtype=torch.float32, device=x.device).contiguous()
        windows = sliding_windows(h, w, size, stride)

        def accumulate(y: torch.Tensor, hi: int, wi: int) -> None:
            ops.launch(ops.make_tile_accumulate(tile=y.contiguous().float(), wts=wts, out=out, count=count, B=b, C=c, H=h,
                                                W=w, th=size, tw=size, hi=hi, wi=wi))

        if batched_fn is not None and len(windows) > 1:
            per_call = max(1, max_batch // b)
            for g0 in range(0, len(windows), per_call):
                group = windows[g0:g0 + per_call]
                xs = torch.cat([x[..., hi:he, wi:we] for hi, he, wi, we in group], dim=0)
                ys = batched_fn(xs, group, *args, **kwargs)
                for k, (hi, he, wi, we) in enumerate(group):
                    accumulate(ys[k * b:(k + 1) * b], hi, wi)
        else:
            for hi, hi_end, wi, wi_end in windows:
                x_tile = x[..., hi:hi_end, wi:wi_end]
                if len(args) or len(kwargs):
                    kwargs.update(dict(hi=hi, hi_end=hi_end, wi=wi, wi_end=wi_end))
                accumulate(fn(x_tile, *args, **kwargs), hi, wi)
        res = torch.empty_like(out)
        ops.launch(ops.make_divide(num=out, den=count, out=res, n=out.numel()))
        return res

    return tiled_fn
