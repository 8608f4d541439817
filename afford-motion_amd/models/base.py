"""`from models.base import create_model_and_diffusion` (reference test.py:8, train.py) -> afm.base."""
from afm.base import Model, create_gaussian_diffusion, create_model, create_model_and_diffusion  # noqa: F401
