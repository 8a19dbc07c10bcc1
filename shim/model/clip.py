"""reference model/clip.py:12 -> edtr_amd."""
from edtr_amd.model.clip import FrozenOpenCLIPEmbedder  # noqa: F401
