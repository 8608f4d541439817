"""Stand-in for omegaconf (absent offline). The reference only uses DictConfig
as a type annotation (models/base.py:3); configs are passed as attribute dicts."""


class DictConfig(dict):
    pass


class OmegaConf:
    """Only what importing the reference's entry scripts touches (test.py:3,24; train.py:6)."""

    @staticmethod
    def to_yaml(cfg) -> str:
        return repr(cfg)

    @staticmethod
    def register_new_resolver(*args, **kwargs) -> None:
        return None
