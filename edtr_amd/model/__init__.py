from .cldm import ControlLDM, ControlNet, ControlledUnetModel, AutoencoderKL  # noqa: F401
from .clip import FrozenOpenCLIPEmbedder  # noqa: F401
from .swinir import SwinIR  # noqa: F401
