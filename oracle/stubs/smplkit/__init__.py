"""Stand-in for smplkit (absent offline). The reference instantiates an
SMPLXLayer at import of utils/misc.py:24; nothing on the denoising path uses it."""
from . import constants  # noqa: F401


class SMPLXLayer:
    def __init__(self, *a, **k):
        pass

    def to(self, *a, **k):
        return self

    def __call__(self, *a, **k):
        raise RuntimeError("smplkit stub: body model is not on the denoising path")
