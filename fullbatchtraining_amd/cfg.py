"""Minimal composer for the reference's Hydra cfg surface (no hydra/omegaconf dependency).

The reference reads one OmegaConf tree ``cfg.{data,model,impl,hyp,analysis,seed,name,dryrun}`` built by Hydra
from ``config/cfg.yaml`` + group files (reference ``config/cfg.yaml:9-37``, ``train_with_gradient_descent.py:19``).
This module composes the same tree from ``fullbatchtraining_amd/config/presets.yaml`` (one document: group -> choice ->
values) and accepts the same command-line override grammar for the keys the hot path consumes:

    compose(["hyp=gradreg", "data.batch_size=32", "hyp.grad_reg.block_strength=0.5"])

Group selection (``hyp=fb1``), dotted value overrides and nested ``defaults:`` lists (with ``_self_`` last
semantics as in Hydra >= 1.1) are supported. Values are parsed as YAML scalars.
"""
import copy
import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config")
_PRESETS = None


def _presets():
    """group path ("root", "hyp", "hyp/optim", ...) -> choice -> values."""
    global _PRESETS
    if _PRESETS is None:
        with open(os.path.join(CONFIG_DIR, "presets.yaml")) as handle:
            _PRESETS = yaml.safe_load(handle)["groups"]
    return _PRESETS


class AttrDict(dict):
    """dict with attribute access; supports ``.items()``, ``**cfg.hyp.grad_reg`` and ``getattr`` like DictConfig."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    def __setattr__(self, key, value):
        self[key] = value

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_attrdict(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attrdict(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [to_attrdict(v) for v in obj]
    return obj


def _parse_scalar(text):
    value = yaml.safe_load(text)
    # YAML 1.1 (PyYAML) reads "1e-2" as a string; OmegaConf reads it as a float.
    if isinstance(value, str):
        try:
            return float(value)
        except ValueError:
            return value
    return value


def _fix_floats(obj):
    """PyYAML parses exponents without a dot (``1e-2``, ``5e-4``) as str; convert those to float like OmegaConf."""
    if isinstance(obj, dict):
        return {k: _fix_floats(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_fix_floats(v) for v in obj]
    if isinstance(obj, str):
        try:
            return float(obj) if any(c in obj for c in "eE") and obj.strip()[0] in "+-.0123456789" else obj
        except ValueError:
            return obj
    return obj


def _merge(base, new):
    for key, value in new.items():
        if isinstance(value, dict) and isinstance(base.get(key), dict):
            _merge(base[key], value)
        else:
            base[key] = copy.deepcopy(value)
    return base


def _load_group(rel_dir, name, selections, prefix):
    """Load choice ``name`` of group ``rel_dir`` resolving its own ``defaults`` list relative to that group."""
    group = _presets().get(rel_dir.replace(os.sep, "/") or "root", {})
    if name not in group:
        raise ValueError(f"Unknown config option {os.path.join(rel_dir, name)!r}.")
    raw = _fix_floats(copy.deepcopy(group[name]) or {})
    defaults = raw.pop("defaults", [])
    out = {}
    for entry in defaults:
        if entry == "_self_":
            continue
        if isinstance(entry, str):  # sibling file in the same group, merged at this level
            _merge(out, _load_group(rel_dir, entry, selections, prefix))
        else:
            ((group, choice),) = entry.items()
            key = f"{prefix}.{group}" if prefix else group
            choice = selections.get(key, choice)
            out[group] = _load_group(os.path.join(rel_dir, group), choice, selections, key)
    _merge(out, raw)
    return out


def compose(overrides=(), **extra):
    """Build the cfg tree. ``overrides`` follow Hydra's CLI grammar (``group=choice`` or ``a.b.c=value``)."""
    selections, assignments = {}, []
    for item in overrides:
        key, _, value = item.partition("=")
        key = key.lstrip("+").replace("/", ".")
        group = _presets().get(key.replace(".", "/"))
        if group is not None and value in group:
            selections[key] = value
        elif group is not None and key.replace(".", "/") != "root":
            raise ValueError(f"Unknown choice {value!r} for config group {key!r} (have: {sorted(group)}).")
        else:
            assignments.append((key, _parse_scalar(value)))
    tree = _load_group("", "cfg", selections, "")
    for key, value in assignments:
        node = tree
        parts = key.split(".")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = value
    tree.update(extra)
    cfg = to_attrdict(tree)
    cfg.setdefault("original_cwd", os.getcwd())
    cfg.setdefault("job_logging_cfg", {"version": 1, "disable_existing_loggers": False})
    return cfg
