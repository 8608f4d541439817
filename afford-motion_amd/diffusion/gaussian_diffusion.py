"""`from diffusion import gaussian_diffusion as gd` (reference models/base.py:29) -> afm.diffusion."""
from afm.diffusion import (GaussianDiffusion, LossType, ModelMeanType, ModelVarType,  # noqa: F401
                           betas_for_alpha_bar, get_named_beta_schedule)

from afm._shim import reference_fallback  # noqa: E402

__getattr__ = reference_fallback(__name__, __file__, allow=())      # every name of this file is on the hot path: nothing falls through
