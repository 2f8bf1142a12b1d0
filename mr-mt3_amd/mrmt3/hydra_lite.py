"""Minimal Hydra-compatible composer used when hydra-core / omegaconf are not installed (they are
absent from the MI355X image).  It consumes the reference's YAML files UNCHANGED
(`config/*.yaml`, `config/model/*.yaml`, `config/dataset/*.yaml`) and supports exactly the features
those files and the reference's shell scripts use (SURVEY §5 "Config / flags"):

  * a `defaults:` list of `- group: option` entries -> merges `<config_dir>/<group>/<option>.yaml`
    under the key `group`;
  * `${a.b}` interpolation against the root, `${hydra:runtime.choices.<group>}`;
  * CLI overrides `key=value`, `a.b=value`, `+new.key=value`, `group=option`, list values `[0,1]`;
  * `_target_` instantiation (`hydra.utils.instantiate` / `get_class` / `get_method`).
When hydra IS importable the reference's own `@hydra.main` path works as is and this module is unused.
"""
from __future__ import annotations

import importlib
import os
import re

import yaml


class _Loader(yaml.SafeLoader):
    """SafeLoader with YAML-1.2 style floats: `2e-4` / `1e-06` are numbers (as OmegaConf reads them)."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(?:[0-9][0-9_]*\.[0-9_]*(?:[eE][-+]?[0-9]+)?|\.[0-9_]+(?:[eE][-+]?[0-9]+)?|[0-9][0-9_]*[eE][-+]?[0-9]+"
               r"|\.(?:inf|Inf|INF)|\.(?:nan|NaN|NAN))$"),
    list("-+0123456789."))


def _load(text):
    return yaml.load(text, Loader=_Loader)


class Cfg(dict):
    """dict with attribute access (what the tasks use on OmegaConf nodes: `cfg.optim.lr`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _parse_value(s: str):
    try:
        return _load(s)
    except yaml.YAMLError:
        return s


def _set(cfg, dotted, value, create):
    parts = dotted.split(".")
    node = cfg
    for p in parts[:-1]:
        if p not in node or not isinstance(node[p], dict):
            if not create:
                raise KeyError(f"override '{dotted}': '{p}' does not exist (use +{dotted}=...)")
            node[p] = Cfg()
        node = node[p]
    if parts[-1] not in node and not create:
        raise KeyError(f"override '{dotted}' does not exist (use +{dotted}=...)")
    node[parts[-1]] = _wrap(value)


def _get(cfg, dotted):
    node = cfg
    for p in dotted.split("."):
        node = node[p]
    return node


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _resolve(cfg, choices):
    def res_str(s, depth=0):
        if depth > 20:
            raise ValueError("interpolation cycle in " + s)

        def one(m):
            key = m.group(1).strip()
            if key.startswith("hydra:runtime.choices."):
                return str(choices[key.split(".")[-1]])
            return _get(cfg, key)
        m = _INTERP.fullmatch(s)
        if m:                                   # whole-string interpolation keeps the value's type
            v = one(m)
            return res_str(v, depth + 1) if isinstance(v, str) and "${" in v else v
        out = _INTERP.sub(lambda mm: str(one(mm)), s)
        return res_str(out, depth + 1) if "${" in out else out

    def walk(node):
        if isinstance(node, dict):
            for k in list(node):
                node[k] = walk(node[k])
            return node
        if isinstance(node, list):
            return [walk(v) for v in node]
        if isinstance(node, str) and "${" in node:
            try:
                return res_str(node)
            except KeyError:
                # OmegaConf resolves lazily: an interpolation of a key this config does not define only fails when
                # that value is read.  The reference's dataset groups are shared by top-level configs that do not all
                # define every key they interpolate (e.g. ${split_frame_length}); leave such values unresolved.
                return node
        return node
    return walk(cfg)


def compose(config_dir: str, config_name: str = "config", overrides=()):
    with open(os.path.join(config_dir, config_name + ".yaml")) as f:
        root = _load(f.read()) or {}
    defaults = root.pop("defaults", []) or []
    choices = {}
    for d in defaults:
        if isinstance(d, dict):
            for g, opt in d.items():
                choices[g] = opt
    plain = []
    for ov in overrides:
        # `key=value`; an `=` inside the value is written `\=` on the command line (e.g. a checkpoint called
        # `...context\=0.ckpt` in the reference's test.sh)
        k, _, v = ov.replace("\\=", "\0").partition("=")
        k, v = k.replace("\0", "="), v.replace("\0", "=")
        if len(v) >= 2 and v[0] == v[-1] and v[0] in "\"'":
            v = v[1:-1]
        if k.lstrip("+~").startswith("hydra/") or k.lstrip("+~").startswith("hydra."):
            continue                            # Hydra's own groups (job_logging, run dir, ...): nothing to configure here
        if k.lstrip("+") in choices and "." not in k and os.path.isdir(os.path.join(config_dir, k.lstrip("+"))):
            choices[k.lstrip("+")] = v          # group selection, e.g. model=MT3NetSegMemV2WithPrev
        else:
            plain.append((k, v))
    cfg = _wrap(root)
    for g, opt in choices.items():
        with open(os.path.join(config_dir, g, f"{opt}.yaml")) as f:
            cfg[g] = _wrap(_load(f.read()) or {})
    for k, v in plain:
        create = k.startswith("+")
        _set(cfg, k.lstrip("+"), _parse_value(v), create)
    return _resolve(cfg, choices)


def get_object(path: str):
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


get_class = get_method = get_object


def instantiate(node, **kwargs):
    """`hydra.utils.instantiate`: builds `_target_(**rest, **kwargs)`; nested dicts stay configs."""
    node = dict(node)
    target = node.pop("_target_")
    node.update(kwargs)
    return get_object(target)(**node)
