from .cldm import ControlLDM, ControlNet, ControlledUnetModel, AutoencoderKL, PromptEncoder  # noqa: F401
