"""The functions of reference utils/common.py that the restoration scripts call around the hot path
(demo.py:14-21,89-124; main/det/test_edtr.py:10-21), backed by edtr_amd."""
import torch
import torch.nn.functional as F

from edtr_amd.evalutil import calculate_psnr_pt, rgb2ycbcr_pt  # noqa: F401
from edtr_amd.shim import get_obj_from_str, instantiate_from_config  # noqa: F401
from edtr_amd.tiling import gaussian_weights, make_tiled_fn, sliding_windows  # noqa: F401
from edtr_amd.wavelet import wavelet_decomposition, wavelet_reconstruction  # noqa: F401


def pad_if_smaller(imgs: torch.Tensor, size: int) -> torch.Tensor:
    """utils/common.py:337-340."""
    _, _, h, w = imgs.size()
    return F.pad(imgs, pad=(0, max(size - w, 0), 0, max(size - h, 0)), mode="constant", value=0)


def pad_to_multiples_of(imgs: torch.Tensor, multiple: int) -> torch.Tensor:
    """utils/common.py:343-348."""
    _, _, h, w = imgs.size()
    if h % multiple == 0 and w % multiple == 0:
        return imgs.clone()
    ph, pw = ((v + multiple - 1) // multiple * multiple - v for v in (h, w))
    return F.pad(imgs, pad=(0, pw, 0, ph), mode="constant", value=0)
