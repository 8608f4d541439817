"""Deterministic synthetic weights and inputs for the denoising hot path.

No checkpoints or datasets exist offline, so every test,
golden fixture and bench run draws from the generators below.  Everything is
keyed by *name* (state-dict key / input name) and a seed, never by call order,
so the same tensors are reproduced in the golden-generation container, in the
CPU oracle and on the GPU box.

Shapes/ranges follow SURVEY.md §8d: scene chunks are 4 m x 4 m centred clouds
with 60 % floor points (reference: prepare/generate_contact_data.py:401-435),
contact maps live in (0, 1] (datasets/humanml3d.py:773-774), frame masks are
suffix pads with lengths that are multiples of 4 in [40, 196]
(datasets/humanml3d.py:777).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch

DATA_SEED = 2023      # reference default seed, configs/default.yaml:47
WEIGHT_SEED = 1234
NOISE_SEED = 7


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(name.encode()) & 0xFFFFFFFF, seed & 0xFFFFFFFF])


def make_tensor_for(name: str, shape: Tuple[int, ...], seed: int = WEIGHT_SEED) -> torch.Tensor:
    """One parameter/buffer tensor, distribution chosen from the key's suffix.

    * ``running_var``             U(0.5, 1.5)   (kept positive)
    * ``running_mean``            N(0, 0.1)
    * ``num_batches_tracked``     0 (int64)
    * norm / bn ``weight``        1 + N(0, 0.1)
    * ``bias``                    N(0, 0.02)
    * matrices                    N(0, 1/sqrt(fan_in))  (activations stay O(1))
    """
    rng = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        a = rng.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_mean":
        a = rng.normal(0.0, 0.1, size=shape)
    elif len(shape) <= 1 and leaf in ("weight",):
        a = 1.0 + rng.normal(0.0, 0.1, size=shape)
    elif leaf in ("bias", "in_proj_bias"):
        a = rng.normal(0.0, 0.02, size=shape)
    else:
        fan_in = shape[-1] if len(shape) >= 2 else max(1, shape[0])
        a = rng.normal(0.0, 1.0 / np.sqrt(fan_in), size=shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def make_state_dict(shapes: Mapping[str, Tuple[Tuple[int, ...], torch.dtype]] | Iterable,
                    seed: int = WEIGHT_SEED, skip: Tuple[str, ...] = ("pe",)) -> Dict[str, torch.Tensor]:
    """Weights for every (name -> (shape, dtype)) entry; keys whose leaf is in
    ``skip`` (deterministic buffers such as the sinusoid table) are omitted."""
    out: Dict[str, torch.Tensor] = {}
    items = shapes.items() if isinstance(shapes, Mapping) else shapes
    for name, (shape, _dtype) in items:
        if name.rsplit(".", 1)[-1] in skip:
            continue
        out[name] = make_tensor_for(name, tuple(shape), seed)
    return out


def fill_module_(module: torch.nn.Module, seed: int = WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    """Overwrite every parameter/buffer of ``module`` (except sinusoid tables and
    anything under a frozen text/scene model) in place; returns the dict used."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if k.rsplit(".", 1)[-1] == "pe" or "text_model" in k or "clip_model" in k:
            continue
        new[k] = make_tensor_for(k, tuple(v.shape), seed).to(v.dtype)
    module.load_state_dict(new, strict=False)
    return new


# --------------------------------------------------------------------------- inputs

def scene_cloud(batch: int, num_points: int, seed: int = DATA_SEED) -> torch.Tensor:
    """``c_pc_xyz`` [B, N, 3]: x,y ~ U(-2,2); 60 % floor (z = tiny jitter so the
    cloud stays tie-free), 40 % furniture z ~ U(0,2)."""
    rng = _rng("c_pc_xyz", seed)
    xy = rng.uniform(-2.0, 2.0, size=(batch, num_points, 2))
    z = rng.uniform(0.0, 2.0, size=(batch, num_points))
    floor = rng.uniform(0.0, 1.0, size=(batch, num_points)) < 0.6
    z = np.where(floor, rng.uniform(0.0, 1e-3, size=(batch, num_points)), z)
    return torch.from_numpy(np.concatenate([xy, z[..., None]], -1).astype(np.float32))


def contact_map(batch: int, num_points: int, joints: int = 6, seed: int = DATA_SEED) -> torch.Tensor:
    """``c_pc_contact`` [B, N, J] in (0, 1] (range of exp(-d^2 / 2 sigma^2))."""
    rng = _rng("c_pc_contact", seed)
    return torch.from_numpy(rng.uniform(1e-3, 1.0, size=(batch, num_points, joints)).astype(np.float32))


def text_feature(batch: int, dim: int = 512, seed: int = DATA_SEED) -> torch.Tensor:
    """Stand-in for the frozen CLIP pooled text feature [B, dim]."""
    rng = _rng("c_text_feat", seed)
    return torch.from_numpy((rng.normal(0.0, 1.0, size=(batch, dim)) * 0.05).astype(np.float32))


def gaussian(name: str, shape: Tuple[int, ...], seed: int = NOISE_SEED) -> torch.Tensor:
    return torch.from_numpy(_rng(name, seed).normal(0.0, 1.0, size=shape).astype(np.float32))


def frame_mask(batch: int, frames: int, seed: int = DATA_SEED, all_valid: bool = False,
               min_len: int = 40) -> torch.Tensor:
    """``x_mask`` [B, L] bool, True = padded frame (suffix padding)."""
    if all_valid:
        return torch.zeros(batch, frames, dtype=torch.bool)
    rng = _rng("x_mask", seed)
    lo = min(min_len, frames) // 4
    lens = rng.integers(max(1, lo), frames // 4 + 1, size=batch) * 4
    lens = np.minimum(lens, frames)
    ar = np.arange(frames)[None, :]
    return torch.from_numpy(ar >= lens[:, None])
