from afm.cmdm import CMDM  # noqa: F401

__all__ = ["CMDM"]
