"""reference model/controlnet.py:18,44 -> edtr_amd."""
from edtr_amd.model.cldm import ControlledUnetModel, ControlNet  # noqa: F401
