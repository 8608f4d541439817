"""Minimal attribute-dict configuration + YAML loader for the reference's `configs/` key surface.

Hydra / OmegaConf are not available offline; the models only ever do duck-typed attribute
access on the config (reference models/cmdm.py:19-70, models/cdm.py:418-472, models/base.py:32-70),
so a nested attribute dict with `${a.b.c}` interpolation covers the contract.
"""
from __future__ import annotations

import os
import re
from typing import Any, Dict, Iterable, Optional

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")


class Config(dict):
    """dict with attribute access (recursively)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_config(obj: Any) -> Any:
    if isinstance(obj, dict):
        return Config({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [to_config(v) for v in obj]
    return obj


def _lookup(root: Dict, path: str):
    cur = root
    for part in path.split("."):
        cur = cur[part]
    return cur


_PAT = re.compile(r"\$\{([^}:]+)\}")


def _resolve(node: Any, root: Dict, depth: int = 0) -> Any:
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve(node[k], root, depth)
        return node
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str):
        m = _PAT.fullmatch(node)
        if m:
            try:
                return _resolve(_lookup(root, m.group(1)), root, depth + 1) if depth < 8 else node
            except (KeyError, TypeError):
                return node
    return node


def set_by_path(cfg: Dict, dotted: str, value: Any) -> None:
    cur = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        cur = cur.setdefault(p, Config())
    cur[parts[-1]] = value


def load_config(task: Optional[str] = None, model: Optional[str] = None, overrides: Optional[Iterable[str]] = None,
                config_dir: str = CONFIG_DIR) -> Config:
    """Compose default.yaml + task/<task>.yaml + model/<model>.yaml, apply `a.b=c` overrides
    (the CLI form used by the reference's scripts/*/test.sh), then resolve `${...}`."""
    def read(*parts):
        with open(os.path.join(config_dir, *parts)) as f:
            return yaml.safe_load(f) or {}
    cfg = read("default.yaml")
    cfg.pop("hydra", None)
    cfg.pop("defaults", None)
    if task:
        cfg["task"] = read("task", task + ".yaml")
    if model:
        cfg["model"] = read("model", model + ".yaml")
    cfg = to_config(cfg)
    for ov in overrides or ():
        k, v = ov.split("=", 1)
        set_by_path(cfg, k, yaml.safe_load(v))
    return to_config(_resolve(cfg, cfg))
