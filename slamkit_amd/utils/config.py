"""Tiny stand-in for Hydra/OmegaConf (absent here): attribute dicts, YAML files with a `defaults`
list, and `a.b.c=value` command-line overrides - the same keys as /root/reference config/**/*.yaml."""
from __future__ import annotations

import os
from typing import Any, List

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config")


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return super().get(k, default)


def to_config(x):
    if isinstance(x, dict):
        return Config({k: to_config(v) for k, v in x.items()})
    if isinstance(x, list):
        return [to_config(v) for v in x]
    return x


def to_container(x):
    if isinstance(x, dict):
        return {k: to_container(v) for k, v in x.items()}
    if isinstance(x, list):
        return [to_container(v) for v in x]
    return x


def _merge(a: dict, b: dict) -> dict:
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _merge(a[k], v)
        else:
            a[k] = v
    return a


def _load_file(rel: str) -> dict:
    path = os.path.join(CONFIG_DIR, rel if rel.endswith(".yaml") else rel + ".yaml")
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    out: dict = {}
    group = os.path.dirname(rel)
    for d in raw.pop("defaults", []) or []:
        if d == "_self_":
            continue
        if isinstance(d, str):  # same-group default, e.g. "- default"
            _merge(out, _load_file(os.path.join(group, d)))
        else:  # {group: option} -> nested under `group`
            for g, opt in d.items():
                _merge(out, {g: _load_file(os.path.join(g, opt))})
    return _merge(out, raw)


def _parse_value(s: str) -> Any:
    try:
        v = yaml.safe_load(s)
    except Exception:
        return s
    if isinstance(v, str):  # YAML 1.1 does not read "3e-3" as a float
        try:
            return float(v)
        except ValueError:
            return v
    return v


def load_config(name: str, overrides: List[str] = ()) -> Config:
    """`name` = top-level yaml (e.g. 'train'); overrides like `model=slam` (group choice),
    `data.train_path=...`, `training_args.max_steps=10`."""
    cfg = _load_file(name)
    for ov in overrides:
        key, _, val = ov.partition("=")
        key = key.lstrip("+")
        if "." not in key and os.path.isdir(os.path.join(CONFIG_DIR, key)) and \
                os.path.exists(os.path.join(CONFIG_DIR, key, f"{val}.yaml")):
            cfg[key] = _load_file(os.path.join(key, val))
            continue
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _parse_value(val)
    return to_config(cfg)
