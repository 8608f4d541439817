from afm.cmdm import compute_repr_dimesion  # noqa: F401
