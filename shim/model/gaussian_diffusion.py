"""reference model/gaussian_diffusion.py:9-84 -> edtr_amd."""
from edtr_amd.diffusion import Diffusion, extract_into_tensor, make_beta_schedule  # noqa: F401
