"""`from diffusion.respace import SpacedDiffusion, space_timesteps` (reference models/base.py:30) -> afm.diffusion."""
from afm.diffusion import SpacedDiffusion, space_timesteps  # noqa: F401

from afm._shim import reference_fallback  # noqa: E402

__getattr__ = reference_fallback(__name__, __file__, allow=())      # every name of this file is on the hot path: nothing falls through
