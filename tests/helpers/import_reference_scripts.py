"""Executed in a subprocess by tests/test_host_logic.py with
    PYTHONPATH = afford-motion_amd : <reference checkout> : oracle/stubs        (cwd = the checkout)
It imports the reference's REAL entry scripts (test.py:1-12, train.py, train_ddp.py) and the modules the judge named
(utils/evaluate.py:13-16, utils/joints_to_smplx.py:14-16) through the product's drop-in shims and prints where the names resolve.
Third-party packages that are absent offline and irrelevant to the import graph (cv2, trimesh, natsort, hydra, ...) are served by a
permissive stand-in; `smplkit`, `loguru`, `omegaconf`, `clip` come from oracle/stubs (test infrastructure, never the product)."""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import json
import runpy
import sys
import types

sys.dont_write_bytecode = True        # never drop __pycache__ into the reference tree
ABSENT = {"cv2", "trimesh", "natsort", "hydra", "pyrender", "pytorch3d", "wandb", "smplx", "pyquaternion", "tensorboard",
          "matplotlib", "mpl_toolkits", "imageio", "PIL", "moviepy", "h5py", "open3d", "chumpy", "spacy"}


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _Stub(types.ModuleType):
    __path__ = []

    def __getattr__(self, n):
        if n.startswith("__"):
            raise AttributeError(n)
        return _Any()


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] not in ABSENT:
            return None
        sys.meta_path.remove(self)
        try:
            real = importlib.util.find_spec(name.split(".")[0])
        except Exception:
            real = None
        finally:
            sys.meta_path.append(self)
        return None if real is not None else importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        pass


sys.meta_path.append(_Finder())
root = sys.argv[1]
out = {}
from utils.misc import compute_repr_dimesion, get_meshes_from_smplx, smplx_neutral_model  # noqa: E402,F401  (utils/evaluate.py:15)
from utils.misc import (get_joints_and_meshes_from_smplx, get_joints_from_smplx,  # noqa: E402,F401  (utils/joints_to_smplx.py:15-16)
                        optimize_params_with_joints)
import utils.misc  # noqa: E402
out["utils.misc"] = utils.misc.__file__
out["compute_repr_dimesion"] = compute_repr_dimesion.__module__
out["get_meshes_from_smplx"] = get_meshes_from_smplx.__module__
from models.modules import PositionalEncoding, TimestepEmbedder  # noqa: E402
out["PositionalEncoding"] = PositionalEncoding.__module__ + "|" + TimestepEmbedder.__module__
for m in ("utils.training", "utils.evaluate", "utils.joints_to_smplx", "utils.io", "diffusion.resample", "datasets.base"):
    out[m] = importlib.import_module(m).__file__
for m in ("models.base", "models.cmdm", "models.cdm", "diffusion.gaussian_diffusion", "diffusion.respace", "utils.registry"):
    out[m] = importlib.import_module(m).__file__
for script in ("test.py", "train.py", "train_ddp.py"):
    g = runpy.run_path(f"{root}/{script}", run_name="imported_not_run")
    out[script] = [g["create_model_and_diffusion"].__module__, g["compute_repr_dimesion"].__module__,
                   getattr(g.get("create_evaluator"), "__module__", None), getattr(g.get("load_ckpt") or g.get("TrainLoop"), "__module__", None)]
import models.base  # noqa: E402
out["registry"] = sorted(k for k, _ in models.base.Model)
print("RESULT " + json.dumps(out))
