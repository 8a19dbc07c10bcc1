"""reference model/swinir.py:624 -> edtr_amd."""
from edtr_amd.model.swinir import SwinIR  # noqa: F401
