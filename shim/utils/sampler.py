"""reference utils/sampler.py:14-323 -> edtr_amd."""
from edtr_amd.sampler import SpacedSampler, space_timesteps  # noqa: F401
