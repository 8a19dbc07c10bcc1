"""`model` package of the reference (model/__init__.py:1-16), hot-path classes only, backed by edtr_amd."""
from .controlnet import ControlledUnetModel, ControlNet  # noqa: F401
from .vae import AutoencoderKL  # noqa: F401
from .clip import FrozenOpenCLIPEmbedder  # noqa: F401
from .cldm import ControlLDM  # noqa: F401
from .gaussian_diffusion import Diffusion  # noqa: F401
from .swinir import SwinIR  # noqa: F401
