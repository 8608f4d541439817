from afm.cdm import CDM  # noqa: F401

__all__ = ["CDM"]
