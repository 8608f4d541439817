"""CDM / ADM denoiser (Perceiver) - filled in after the CMDM path."""
from __future__ import annotations

import torch.nn as nn

from .base import Model


@Model.register()
class CDM(nn.Module):
    def __init__(self, cfg, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("CDM Perceiver HIP path not built yet")
