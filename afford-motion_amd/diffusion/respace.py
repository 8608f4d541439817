"""`from diffusion.respace import SpacedDiffusion, space_timesteps` (reference models/base.py:30) -> afm.diffusion."""
from afm.diffusion import SpacedDiffusion, space_timesteps  # noqa: F401
