"""reference model/vae.py:681 -> edtr_amd."""
from edtr_amd.model.cldm import AutoencoderKL  # noqa: F401
