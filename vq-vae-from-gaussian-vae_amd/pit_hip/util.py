"""Config factory for the reference's YAMLs (reference pit/util.py:45-62).

``instantiate_from_config({"target": ..., "params": ...})`` with one addition:
dotted names under the reference's ``pit.`` package resolve to their ``pit_hip.``
counterparts, so the shipped ``configs/*.yaml`` instantiate this implementation
unmodified.  ``load_config`` is a small PyYAML loader that resolves the
``${a.b.c}`` node interpolation the configs use (OmegaConf is not required)."""
from __future__ import annotations

import importlib
import re
from typing import Any

_REMAP = (("pit.", "pit_hip."),)


def get_obj_from_str(string: str, reload: bool = False, invalidate_cache: bool = True):
    for old, new in _REMAP:
        if string.startswith(old):
            string = new + string[len(old):]
            break
    module, cls = string.rsplit(".", 1)
    if invalidate_cache:
        importlib.invalidate_caches()
    mod = importlib.import_module(module, package=None)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**(config.get("params", dict()) or dict()))


_INTERP = re.compile(r"^\$\{([^}]+)\}$")


def _resolve(node: Any, root: Any) -> Any:
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.match(node.strip())
        if m:
            cur = root
            for part in m.group(1).split("."):
                cur = cur[part]
            return _resolve(cur, root)
    return node


def load_config(path: str) -> dict:
    import yaml

    with open(path) as f:
        raw = yaml.safe_load(f)
    return _resolve(raw, raw)
