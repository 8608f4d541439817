from afm.registry import Registry  # noqa: F401
