"""Import the real reference (read-only at /root/reference) in THIS container.

Used only by oracle/make_goldens.py and the `-m "not gpu"` cross-check tests
that skip when /root/reference is absent (it never exists on the GPU box).
Stubs for absent third-party modules live in oracle/stubs.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("AFM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


def import_reference():
    """Returns the reference's (models.base, diffusion.gaussian_diffusion) modules."""
    if not reference_available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True          # never drop __pycache__ into the read-only tree
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (REFERENCE_ROOT, os.path.join(here, "stubs")):
        if p not in sys.path:
            sys.path.insert(0, p)
    # a product checkout on sys.path may already own `models`/`diffusion`/`utils`
    for name in [n for n in sys.modules if n.split(".")[0] in ("models", "diffusion", "utils")]:
        f = getattr(sys.modules[name], "__file__", "") or ""
        if not f.startswith(REFERENCE_ROOT):
            del sys.modules[name]
    if "pointops_cuda" not in sys.modules:
        from oracle import pointops_ref
        mod = types.ModuleType("pointops_cuda")
        mod.furthestsampling_cuda = pointops_ref.furthestsampling_cuda
        mod.knnquery_cuda = pointops_ref.knnquery_cuda
        sys.modules["pointops_cuda"] = mod
    # the reference hard-codes CUDA tensor constructors (pointtransformer.py:60, pointops.py:21-22,40-41)
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    # the reference's `diffusion/` and `utils/` have no __init__.py (namespace packages): a regular package of the same
    # name ANYWHERE on sys.path would win, so the product's drop-in shims are hidden while the reference is imported
    hidden = [p for p in sys.path if os.path.isfile(os.path.join(p, "diffusion", "__init__.py"))]
    saved = list(sys.path)
    sys.path[:] = [p for p in sys.path if p not in hidden]
    try:
        import models.base as base              # noqa: E402  (reference's)
        import models                           # noqa: F401  (runs the registry decorators)
        import diffusion.gaussian_diffusion as gd
        import diffusion.respace as rs
    finally:
        sys.path[:] = saved
    for m in (base, gd, rs):
        assert m.__file__.startswith(REFERENCE_ROOT), m.__file__
    return base, gd


class AttrDict(dict):
    """Attribute-access dict standing in for an OmegaConf node."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    return d
