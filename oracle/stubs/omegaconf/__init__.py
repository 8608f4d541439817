"""Stand-in for omegaconf (absent offline). The reference only uses DictConfig
as a type annotation (models/base.py:3); configs are passed as attribute dicts."""


class DictConfig(dict):
    pass
