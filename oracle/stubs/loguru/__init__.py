"""Stand-in for loguru (absent offline): a logger whose every method is a no-op."""


class _Logger:
    def __getattr__(self, _name):
        return lambda *a, **k: None


logger = _Logger()
