"""reference model/cldm.py:17 -> edtr_amd."""
from edtr_amd.model.cldm import ControlLDM, NansException, disabled_train  # noqa: F401
