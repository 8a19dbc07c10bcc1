"""Test infrastructure only: CPU restatement of the reference hot path (see edtr_oracle.py)."""


def flat_sd(sds):
    """{'unet': sd, 'controlnet': sd, 'vae': sd} -> the flat ``{part.key: tensor}`` dict the oracle functions take."""
    return {f"{part}.{k}": v for part, sd in sds.items() for k, v in sd.items()}
