"""Hydra-less loader for the reference's `conf/` tree (hydra/omegaconf are not required at run time).

Supports what the reference's entry points use (SURVEY section 5 "Config / flags"): the `defaults` list of
main_config.yaml with the `experiment` / `dset` groups, `${a.b}` interpolation, dotted CLI overrides
(`experiment=aero_4-16_512_64 dset=4-16 +filename=x.wav lr=1e-4`), attribute access on the result.
"""
import copy
import os
import re

import yaml


class Config(dict):
    """dict with attribute access, nested (like an OmegaConf DictConfig for read access)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(o):
    if isinstance(o, dict):
        return Config({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


# PyYAML reads "1e-3" as a string (YAML 1.1); hydra/omegaconf read it as a float.
_FLOAT = re.compile(r'^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$')


def _coerce(o):
    if isinstance(o, dict):
        return {k: _coerce(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_coerce(v) for v in o]
    if isinstance(o, str) and _FLOAT.match(o) and not o.isdigit():
        return float(o)
    return o


def _lookup(root, path):
    cur = root
    for p in path.split('.'):
        cur = cur[p]
    return cur


_REF = re.compile(r'\$\{([^}]+)\}')


def _resolve(node, root):
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve(node[k], root)
        return node
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _REF.fullmatch(node)
        if m:
            return _resolve(copy.deepcopy(_lookup(root, m.group(1))), root)
        return _REF.sub(lambda mm: str(_resolve(_lookup(root, mm.group(1)), root)), node)
    return node


def _parse_value(s):
    return _coerce(yaml.safe_load(s))


def load_config(conf_dir, overrides=()):
    with open(os.path.join(conf_dir, 'main_config.yaml')) as f:
        main = _coerce(yaml.safe_load(f))
    groups = {}
    for d in main.pop('defaults', []):
        if isinstance(d, dict):
            for k, v in d.items():
                if not k.startswith('override '):
                    groups[k] = v
    plain = []
    for ov in overrides:
        key, _, val = ov.partition('=')
        key = key.lstrip('+')
        if key in groups:
            groups[key] = val
        else:
            plain.append((key, val))
    main.pop('hydra', None)
    cfg = main
    for g, name in groups.items():
        with open(os.path.join(conf_dir, g, f'{name}.yaml')) as f:
            cfg[g] = _coerce(yaml.safe_load(f)) or {}
    for key, val in plain:
        cur = cfg
        parts = key.split('.')
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = _parse_value(val)
    return _wrap(_resolve(cfg, cfg))
