"""Drop-in `models.modules` (reference models/modules.py): the names other reference files import from it.

On the hot path these are the HIP-backed classes of `afm`; `PositionalEncoding` additionally keeps a plain forward because the
reference's evaluation-time SMPL-X fitter (`utils/joints_to_smplx.py:14`, NOT on the denoising path) uses it as a stand-alone torch
module inside its own autograd optimisation.  (The reference's file cannot be imported on an MI355X box: it pulls in the CUDA-only
`pointops_cuda` extension.)"""
import torch

from afm._shim import reference_fallback
from afm.cmdm import PositionalEncoding as _PositionalEncodingBuffers
from afm.cmdm import TimestepEmbedder  # noqa: F401
from afm.scene import SceneMapEncoder, SceneMapEncoderDecoder  # noqa: F401


class PositionalEncoding(_PositionalEncodingBuffers):
    """x [T, B, d] -> dropout(x + pe[:T])  (reference models/modules.py:28-36)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.dropout(x + self.pe[: x.shape[0], :].to(x))


def get_positional_encoding(max_len: int, time_emb_dim: int) -> torch.Tensor:
    """[max_len, 1, d] sinusoid table (reference models/modules.py:10-25)."""
    from afm.cmdm import sinusoid_table
    return sinusoid_table(max_len, time_emb_dim)


# the Perceiver building blocks that only the reference's own models/cdm.py imports (ours is afm.cdm) -> the checkout's file; every other
# missing name raises
__getattr__ = reference_fallback(__name__, __file__, allow=("CrossAttentionLayer", "SelfAttentionBlock", "SelfAttentionLayer", "CrossAttention",
                                                            "SelfAttention", "MultiHeadAttention", "AbstractAttentionLayer", "Residual",
                                                            "MLP", "ModuleOutput", "RotaryPositionEmbedding", "KVCache"))
