"""On-disk hand-off formats of the reference's two-stage evaluation, so that samples produced
here can be consumed by the reference's own datasets / evaluators and vice versa.

* ADM output  -> ``{save_dir}/H3D/pred_contact/{name}-{caption_index}.npy``: float32 distance maps
  ``sqrt(-2 ln(clip(sample*std+mean, 1e-20, 1)) sigma^2)`` of shape ``(k or 1, N, J)``
  (utils/evaluate.py:41-82, datasets/humanml3d.py:494-511); ``use_raw_dist`` stores the clipped raw distances.
* AMDM input  <- the same file, turned back into contact ``exp(-d^2 / (2 sigma^2))`` (datasets/humanml3d.py:763-774).
* AMDM output -> ``{save_dir}/humanml/{name}-{caption_index}.pkl``: ``{name, text, tokens, motion, m_len}``
  (utils/evaluate.py:100-141); single samples are stored denormalised, k-sample stacks as generated.

Host-side numpy only (file I/O is not on the device path).
"""
from __future__ import annotations

import os
import pickle
from typing import Any, Dict, Sequence

import numpy as np


def _np(a) -> np.ndarray:
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def contact_to_dist(sample, mean=0.0, std=1.0, sigma: float = 0.8, use_raw_dist: bool = False) -> np.ndarray:
    """Normalised ADM sample -> what the evaluator stores (denormalize(clip=True) then the distance transform)."""
    c = _np(sample).astype(np.float32) * np.float32(std) + np.float32(mean)
    if use_raw_dist:
        return c.clip(0.0, None)
    c = c.clip(1e-20, 1.0)
    return np.sqrt(-2 * np.log(c) * sigma ** 2)


def dist_to_contact(dist, sigma: float = 0.8, use_raw_dist: bool = False) -> np.ndarray:
    """Stored distance map -> the `c_pc_contact` condition of the AMDM (datasets/humanml3d.py:773-774)."""
    d = _np(dist)
    return d if use_raw_dist else np.exp(-0.5 * d ** 2 / sigma ** 2)


def save_pred_contact(save_dir: str, name: str, caption_index: Any, sample, *, mean=0.0, std=1.0, sigma: float = 0.8,
                      use_raw_dist: bool = False) -> str:
    """sample [N, J] (one sample, stored as (1, N, J)) or [k, N, J] (k samples) in the model's normalised space."""
    dist = contact_to_dist(sample, mean, std, sigma, use_raw_dist)
    if dist.ndim == 2:
        dist = dist[None, ...]
    path = os.path.join(save_dir, f"H3D/pred_contact/{name}-{caption_index}.npy")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.save(path, dist)
    return path


def load_pred_contact(contact_folder: str, name: str, caption_index: Any, *, sigma: float = 0.8, use_raw_dist: bool = False) -> np.ndarray:
    """-> contact condition (k, N, J) for the AMDM, exactly as ContactMotionHumanML3DDataset.__getitem__ builds it."""
    d = np.load(os.path.join(contact_folder, f"H3D/pred_contact/{name.split('_')[-1]}-{caption_index}.npy"))
    return dist_to_contact(d, sigma, use_raw_dist)


def save_motion_sample(save_dir: str, name: str, caption_index: Any, *, text: str, tokens: Sequence[str], motion, x_mask,
                       mean=None, std=None) -> str:
    """motion [L, D] (denormalised with mean/std when given, as for single samples) or [k, L, D] (stored as generated)."""
    m = _np(motion)
    if mean is not None:
        m = m * _np(std) + _np(mean)
    m_len = (~_np(x_mask).astype(bool)).sum()
    path = os.path.join(save_dir, f"humanml/{name}-{caption_index}.pkl")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as fp:
        pickle.dump({"name": name, "text": text, "tokens": tokens, "motion": m, "m_len": m_len}, fp)
    return path


def load_motion_sample(path: str) -> Dict[str, Any]:
    with open(path, "rb") as fp:
        return pickle.load(fp)
