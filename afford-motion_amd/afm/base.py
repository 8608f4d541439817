"""Model registry + factories: the Python face of the drop-in boundary (reference models/base.py:7-83).
`create_model_and_diffusion(cfg, device=...)` reads exactly the config keys the reference reads."""
from __future__ import annotations

import torch.nn as nn

from .registry import Registry

Model = Registry("model")


def create_model(cfg, *args, **kwargs) -> nn.Module:
    """`Model.get(cfg.model.name)(cfg.model, *args, **kwargs)` (reference models/base.py:9-18)."""
    from . import cdm, cmdm  # noqa: F401  (registration by import side effect, like the reference's models/__init__.py)
    return Model.get(cfg.model.name)(cfg.model, *args, **kwargs)


def create_gaussian_diffusion(cfg, *args, **kwargs):
    """SpacedDiffusion from `cfg.diffusion.{steps,noise_schedule,timestep_respacing,predict_xstart,loss_type,
    learn_sigma,sigma_small,rescale_timesteps}` (reference models/base.py:20-70)."""
    from . import diffusion as gd
    c = cfg.diffusion
    steps = c.steps
    respacing = c.timestep_respacing if c.timestep_respacing else [steps]
    mean_type = gd.ModelMeanType.START_X if c.predict_xstart else gd.ModelMeanType.EPSILON
    loss_type = {"MSE": gd.LossType.MSE, "RESCALED_MSE": gd.LossType.RESCALED_MSE, "KL": gd.LossType.KL,
                 "RESCALED_KL": gd.LossType.RESCALED_KL}[c.loss_type]
    if c.learn_sigma:
        var_type = gd.ModelVarType.LEARNED_RANGE
    else:
        var_type = gd.ModelVarType.FIXED_SMALL if c.sigma_small else gd.ModelVarType.FIXED_LARGE
    return gd.SpacedDiffusion(use_timesteps=gd.space_timesteps(steps, respacing),
                              betas=gd.get_named_beta_schedule(c.noise_schedule, steps),
                              model_mean_type=mean_type, model_var_type=var_type, loss_type=loss_type,
                              rescale_timesteps=c.rescale_timesteps)


def create_model_and_diffusion(cfg, *args, **kwargs):
    return create_model(cfg, *args, **kwargs), create_gaussian_diffusion(cfg, *args, **kwargs)
