"""Deterministic fake of openai/CLIP (package + weights absent offline).

Signatures follow the reference call sites (models/functions.py:55,77,83):
``load(version, device, jit) -> (model, preprocess)``,
``tokenize(texts, context_length=77, truncate=False) -> int64 [B, context_length]``,
``model.encode_text(tokens[B,77]) -> float [B, 512]``.
The text feature is a boundary *input* of the denoising path (SURVEY.md §2 row
10), so the fake only has to be deterministic: a table-lookup mean over tokens.
"""
import zlib

import numpy as np
import torch

_DIM = {"ViT-B/32": 512, "ViT-L/14@336px": 768}
_VOCAB = 4096


def tokenize(texts, context_length: int = 77, truncate: bool = False):
    if isinstance(texts, str):
        texts = [texts]
    out = torch.zeros(len(texts), context_length, dtype=torch.int64)
    for i, t in enumerate(texts):
        ids = [1] + [2 + (zlib.crc32(w.encode()) % (_VOCAB - 3)) for w in t.lower().split()]
        ids = ids[: context_length - 1] + [_VOCAB - 1]
        out[i, : len(ids)] = torch.tensor(ids)
    return out


class _FakeClip(torch.nn.Module):
    def __init__(self, dim):
        super().__init__()
        rng = np.random.default_rng(20240101)
        tab = (rng.normal(0, 1, size=(_VOCAB, dim)) * 0.2).astype(np.float32)
        tab[0] = 0
        self.register_buffer("table", torch.from_numpy(tab))

    def encode_text(self, tokens):
        e = self.table.to(tokens.device)[tokens]                # [B, 77, D]
        n = (tokens != 0).sum(-1, keepdim=True).clamp(min=1)
        return e.sum(1) / n


def load(version, device="cpu", jit=False):
    return _FakeClip(_DIM[version]).to(device), None
