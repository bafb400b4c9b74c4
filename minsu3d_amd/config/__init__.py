"""Tiny Hydra-shaped config loader (the reference uses Hydra/OmegaConf, which are not in this image):
`defaults` lists, group selection (`model=hais`), dotted `key=value` overrides, `${a.b}` interpolation and
attribute access over the defaults in config/defaults.py."""
import copy
import re

import yaml

from .defaults import GROUPS, TOP


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return Cfg({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def _merge(a, b):
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            _merge(a[k], v)
        else:
            a[k] = v
    return a


def _load_group(group, name):
    d = copy.deepcopy(GROUPS[group][name])
    out = {}
    for base in d.pop("defaults", []):
        _merge(out, _load_group(group, base))
    return _merge(out, d)


def _resolve(root, node):
    pat = re.compile(r"\$\{([^}]+)\}")

    def lookup(path):
        cur = root
        for p in path.split("."):
            cur = cur[p]
        return cur

    def res(v):
        if isinstance(v, str):
            for _ in range(8):
                m = pat.search(v)
                if not m:
                    break
                val = lookup(m.group(1))
                v = val if m.group(0) == v else v.replace(m.group(0), str(val))
                if not isinstance(v, str):
                    break
            return v
        if isinstance(v, dict):
            return {k: res(x) for k, x in v.items()}
        if isinstance(v, list):
            return [res(x) for x in v]
        return v

    return res(node)


def load_config(overrides=()):
    """load_config(["model=hais", "model.network.m=16", "data.batch_size=2"])"""
    top = copy.deepcopy(TOP)
    groups = {}
    for item in top.pop("defaults", []):
        (g, n), = item.items()
        groups[g] = n
    dotted = []
    for ov in overrides:
        k, v = ov.split("=", 1)
        if k in groups:
            groups[k] = v
        else:
            dotted.append((k, yaml.safe_load(v)))
    cfg = dict(top)
    for g, n in groups.items():
        cfg[g] = _load_group(g, n)
    for k, v in dotted:
        cur = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = v
    return _wrap(_resolve(cfg, cfg))
