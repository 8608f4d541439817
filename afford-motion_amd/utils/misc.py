"""Drop-in `utils.misc` (reference utils/misc.py).  The hot path needs `compute_repr_dimesion` (`models/cmdm.py:10`, `test.py:12`,
`train.py:13`, `datasets/humanml3d.py:14`); every other name of the reference's file (`smplx_neutral_model`,
`get_meshes_from_smplx` for `utils/evaluate.py:15`; `optimize_params_with_joints`, `get_joints_from_smplx`, ... for
`utils/joints_to_smplx.py:15-16`) is served lazily from the reference checkout's own utils/misc.py that follows on sys.path."""
from afm._shim import reference_fallback
from afm.cmdm import compute_repr_dimesion  # noqa: F401

# the SMPL-X helpers (evaluation / visualisation only; none is on the denoising path) - nothing else falls through
__getattr__ = reference_fallback(__name__, __file__, allow=("smplx_neutral_model", "get_meshes_from_smplx", "get_joints_from_smplx",
                                                            "get_joints_and_meshes_from_smplx", "optimize_params_with_joints",
                                                            "compute_optimization_loss"))
