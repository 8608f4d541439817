"""Drop-in `utils.registry` (reference utils/registry.py:10-92): `Registry` is ours (same observable behaviour) and the only name the
reference's file defines - nothing falls through to a checkout."""
from afm._shim import reference_fallback
from afm.registry import Registry  # noqa: F401

__getattr__ = reference_fallback(__name__, __file__, allow=())
