"""Fall-through for the drop-in shim modules (`utils.misc`, `utils.registry`, `models.modules`, ...).

A shim module shadows the reference's file of the same dotted name so that the hot-path names resolve to the HIP-backed
classes; the reference's own scripts, however, import OTHER names from those files as well
(`utils/evaluate.py:15` wants `smplx_neutral_model, get_meshes_from_smplx` from `utils.misc`; `utils/joints_to_smplx.py:15-16`
wants `optimize_params_with_joints, get_joints_from_smplx, ...`).  `reference_fallback()` gives the shim a PEP-562 module
`__getattr__`: a name the shim does not define AND that is on the shim's explicit allow-list is looked up in the same-named file
of the reference checkout that follows on `sys.path` (found through the parent package's extended `__path__`), loaded lazily, once,
under a private module name.  Names the shim defines always win; nothing is loaded unless an allowed missing name is asked for, so a
box without a checkout (or without the checkout's third-party deps, e.g. `smplkit`) can still import every shim.

The allow-list is per shim and holds only names that are NOT on the denoising path (the SMPL-X helpers of `utils.misc`, the Perceiver
building blocks the reference's own `models/cdm.py` imports from `models.modules`).  Every other missing name raises `AttributeError`
naming the shim: a typo or a hot-path name the shim forgot (`diffusion.gaussian_diffusion._extract_into_tensor`, `respace._WrappedModel`)
must fail loudly instead of silently executing the reference's code.
"""
from __future__ import annotations

import importlib.util
import os
import sys
from typing import Callable, Iterable, Optional


def _reference_file(shim_name: str, shim_file: str) -> Optional[str]:
    pkg_name, _, leaf = shim_name.rpartition(".")
    pkg = sys.modules.get(pkg_name)
    own = os.path.abspath(shim_file)
    for d in list(getattr(pkg, "__path__", []) or []):
        for cand in (os.path.join(d, leaf + ".py"), os.path.join(d, leaf, "__init__.py")):
            cand = os.path.abspath(cand)
            if cand != own and os.path.isfile(cand):
                return cand
    return None


def reference_fallback(shim_name: str, shim_file: str, allow: Iterable[str] = ()) -> Callable[[str], object]:
    """Returns a module-level `__getattr__` for the shim module `shim_name` (its `__name__`) at `shim_file`; only the names in
    `allow` may resolve to the reference checkout's file."""
    state = {}
    allowed = frozenset(allow)

    def _load():
        if "mod" in state:
            return state["mod"]
        path = _reference_file(shim_name, shim_file)
        if path is None:
            state["mod"] = None
            return None
        private = shim_name.rpartition(".")[0] + "._reference_" + shim_name.rpartition(".")[2]
        spec = importlib.util.spec_from_file_location(private, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[private] = mod
        try:
            spec.loader.exec_module(mod)
        except BaseException:
            sys.modules.pop(private, None)
            raise
        state["mod"] = mod
        return mod

    def __getattr__(name: str):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if name not in allowed:
            raise AttributeError(
                f"module '{shim_name}' (afford-motion_amd drop-in shim) has no attribute '{name}': the shim does not define it and it is "
                f"not on the shim's allow-list of non-hot-path names served from a reference checkout ({sorted(allowed) or 'none'})")
        mod = _load()
        if mod is None:
            raise AttributeError(
                f"module '{shim_name}' (afford-motion_amd drop-in shim) has no attribute '{name}', and no reference checkout "
                f"providing {shim_name.replace('.', '/')}.py follows it on sys.path")
        try:
            return getattr(mod, name)
        except AttributeError:
            raise AttributeError(f"neither the afford-motion_amd shim '{shim_name}' nor the reference's "
                                 f"{mod.__file__} defines '{name}'") from None

    return __getattr__
