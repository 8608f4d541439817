"""Text-condition boundary.  The frozen CLIP/BERT text encoder is a third-party model outside the
hot path; its pooled feature [B, 512] enters the path as a tensor.

Resolution order for a forward call:
  1. ``c_text_feat`` in the kwargs (tensor [B, text_dim])            - our extension, used by tests/bench;
  2. ``model.text_encoder`` (callable ``list[str] -> [B, text_dim]``) - pluggable;
  3. the `clip` package, loaded like the reference does (models/functions.py:42-84) when importable.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


def lang_feat_dim_type(model_name: str):
    """(feature dim, family) per text model name (reference models/functions.py:86-94)."""
    table = {"bert-base-uncased": (768, "bert"), "ViT-B/32": (512, "clip"), "ViT-L/14@336px": (768, "clip")}
    if model_name not in table:
        raise NotImplementedError(model_name)
    return table[model_name]


class TextEncoderMixin:
    text_encoder: Optional[Callable] = None

    def _init_text_encoder(self) -> None:
        if self.text_feat_type not in ("clip", "bert"):
            raise NotImplementedError(self.text_feat_type)
        self.text_encoder = None
        object.__setattr__(self, "_clip", None)

    def _clip_encode(self, texts, device):
        if self._clip is None:
            try:
                import clip                       # third-party, absent offline
            except ImportError as e:
                raise RuntimeError(
                    "no text encoder available: pass `c_text_feat` ([B, text_dim] pooled text feature), set "
                    "`model.text_encoder`, or install openai/CLIP (the reference's frozen text model)") from e
            model, _ = clip.load(self.text_model_name, device="cpu", jit=False)
            model.eval().requires_grad_(False)
            # NOT a submodule: plain attribute writes bypass nn.Module.__setattr__, so the frozen text model never enters
            # state_dict() / parameters() (the reference keeps it out of checkpoints by key filter, utils/training.py:97)
            object.__setattr__(self, "_clip_mod", clip)
            object.__setattr__(self, "_clip", model)
        clip = self._clip_mod
        ctx = self.text_max_length + 2          # reference models/functions.py:73-79: truncate, then zero-pad to 77
        tok = clip.tokenize(texts, context_length=ctx, truncate=True)
        tok = torch.cat([tok, torch.zeros(tok.shape[0], 77 - ctx, dtype=tok.dtype)], dim=1)
        return self._clip.to(device).encode_text(tok.to(device)).detach().float()

    def encode_text(self, kwargs) -> torch.Tensor:
        if "c_text_feat" in kwargs:
            return kwargs["c_text_feat"].float()
        texts = kwargs["c_text"]
        device = next(self.parameters()).device
        if self.text_encoder is not None:
            return self.text_encoder(texts).to(device).float()
        if self.text_feat_type == "clip":
            return self._clip_encode(texts, device)
        raise NotImplementedError("BERT text features: pass `c_text_feat` or set `model.text_encoder`")
