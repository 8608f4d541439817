"""Two-stage sampling ADM (contact map over the scene) -> AMDM (motion) in ONE process.

The reference runs the stages as two `test.py` invocations that communicate through files:
`ContactHumanML3DEvaluator.evaluate` writes `H3D/pred_contact/{name}-{caption}.npy` = sqrt(-2 ln(c) sigma^2)
(utils/evaluate.py:41-82) and `ContactMotionHumanML3DDataset.__getitem__` reads it back and applies
exp(-d^2 / (2 sigma^2)) (datasets/humanml3d.py:763-774), looping `for k in range(k_samples)` sequentially
(test.py:88-101).  Here the k samples are the batch dimension, the hand-off stays in HBM
(`afm.dist.adm_to_amdm_condition`), and with several ranks the batch is sharded with one gather at the end.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import dist as adist


def two_stage_sample(adm, adm_diffusion, amdm, amdm_diffusion, *, text_feat: torch.Tensor, xyz: torch.Tensor, frames: int,
                     x_mask: Optional[torch.Tensor] = None, sigma: float = 0.8, contact_mean: float = 0.0,
                     contact_std: float = 1.0, seed: int = 0, sample_index0: int = 0,
                     adm_noise: Optional[Dict[str, torch.Tensor]] = None,
                     amdm_noise: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """text_feat [B, text_dim], xyz [B, N, 3] (B = scenes x k_sample, already flattened) ->
    {"contact": [B, N, J] ADM sample, "cond": [B, N, J] AMDM condition, "motion": [B, frames, D]}.

    ``*_noise`` = optional {"x_T": ..., "steps": [T, ...]} explicit noise (parity tests); otherwise Philox
    keyed by (seed, sample_index0 + b), stage 2 uses seed + 1."""
    B, N = xyz.shape[0], xyz.shape[1]
    dev = xyz.device
    adm_kw = dict(c_text_feat=text_feat, c_pc_xyz=xyz)
    an = adm_noise or {}
    contact = adm_diffusion.p_sample_loop(adm, (B, N, adm.contact_dim), noise=an.get("x_T"), clip_denoised=False,
                                          model_kwargs=adm_kw, step_noise=an.get("steps"), seed=seed,
                                          sample_index0=sample_index0)
    cond = adist.adm_to_amdm_condition(contact, sigma=sigma, mean=contact_mean, std=contact_std)
    if x_mask is None:
        x_mask = torch.zeros(B, frames, dtype=torch.bool, device=dev)
    amdm_kw = dict(c_text_feat=text_feat, c_pc_xyz=xyz, c_pc_contact=cond, x_mask=x_mask)
    mn = amdm_noise or {}
    motion = amdm_diffusion.p_sample_loop(amdm, (B, frames, amdm.motion_dim), noise=mn.get("x_T"), clip_denoised=False,
                                          model_kwargs=amdm_kw, step_noise=mn.get("steps"), seed=seed + 1,
                                          sample_index0=sample_index0)
    return {"contact": contact, "cond": cond, "motion": motion}
