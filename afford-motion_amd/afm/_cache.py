"""Keys for the step-invariant caches (condition tokens, scene features, trans_dec memories, text latents).

A cache entry is reused while the caller passes the SAME tensors again (every step of a sampling loop passes the same batch dict).
"Same" = same address, shape, strides, dtype and in-place version counter, AND the entry holds a reference to the keyed tensors:
without that reference the caching allocator hands a freed batch's address to the next batch (same shape, `_version` 0 again) and
the stale entry would match - the address comparison is only meaningful while the old storage is provably still alive."""
from __future__ import annotations

from typing import Any, Optional, Sequence, Tuple

import torch


def _sig(t) -> Optional[tuple]:
    if not isinstance(t, torch.Tensor):
        return None
    return (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype, str(t.device))


class HeldKey:
    """Signature of a tuple of (optional) tensors plus hashable extras; keeps the tensors alive for as long as the key lives."""

    __slots__ = ("sig", "extra", "held")

    def __init__(self, tensors: Sequence[Any], extra: Tuple = ()) -> None:
        self.sig = tuple(_sig(t) for t in tensors)
        self.extra = extra
        self.held = tuple(t for t in tensors if isinstance(t, torch.Tensor))

    def matches(self, tensors: Sequence[Any], extra: Tuple = ()) -> bool:
        return self.extra == extra and self.sig == tuple(_sig(t) for t in tensors)
