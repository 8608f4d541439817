"""Name -> class registry: the plugin surface the reference's entry points rely on
(`Model.get(cfg.model.name)(cfg.model, device=...)`, reference models/base.py:7,18 and
utils/registry.py:10-92).  Same observable behaviour: duplicate registration is an
AssertionError, unknown names a KeyError, `register()` works as decorator or call."""
from __future__ import annotations

from typing import Any, Dict, Iterator, Tuple


class Registry:
    def __init__(self, name: str) -> None:
        self._name = name
        self._obj_map: Dict[str, Any] = {}

    def _add(self, obj: Any) -> Any:
        key = obj.__name__
        assert key not in self._obj_map, f"An object named '{key}' was already registered in '{self._name}' registry!"
        self._obj_map[key] = obj
        return obj

    def register(self, obj: Any = None) -> Any:
        if obj is None:
            return self._add              # used as @REG.register()
        self._add(obj)                    # used as REG.register(cls)

    def get(self, name: str) -> Any:
        if name not in self._obj_map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self._obj_map[name]

    def __contains__(self, name: str) -> bool:
        return name in self._obj_map

    def __iter__(self) -> Iterator[Tuple[str, Any]]:
        return iter(self._obj_map.items())

    def __repr__(self) -> str:
        rows = "\n".join(f"  {k}: {v}" for k, v in self._obj_map.items())
        return f"Registry of {self._name}:\n{rows}"

    __str__ = __repr__
