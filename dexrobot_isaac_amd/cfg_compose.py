"""Hydra-free composer for the reference's configuration tree.

The reference composes `dexhand_env/cfg/{config,task/*,physics/*,train/*,base/*}.yaml` with Hydra 1.2 and hands the env
a plain resolved dict (reference train.py:246-253).  Hydra/omegaconf are not installed here, so this module implements
the subset of the semantics that tree uses, on the reference's YAML files unchanged:

  * `defaults:` lists with `_self_`, same-group names (`- BaseTask`), absolute entries (`- /physics/default`),
    group selections (`- task: BaseTask`) and nested paths (`- base/video`);
  * `# @package _global_ | <path>` headers (default package = the config group);
  * dotted command-line overrides: `task=BlindGrasping` (group choice), `env.numEnvs=2048`, `+a.b=c`;
  * `${a.b}` interpolation (a value that interpolates its own key keeps the value it overrides, which is how
    `env.numEnvs: ${env.numEnvs}` in task/BlindGrasping.yaml behaves once a CLI override is present);
  * `_delete_: true` is kept as an ordinary key, exactly as OmegaConf does (the reference never acts on it).

    cfg = compose("/path/to/dexhand_env/cfg", overrides=["task=BlindGrasping", "env.numEnvs=4096"])
    env = make_env("BlindGrasping", 4096, "cuda:0", "cuda:0", 0, cfg=cfg)
"""
import copy
import os
import re

import yaml

_PKG = re.compile(r"^#\s*@package\s+(\S+)")
_INTERP = re.compile(r"^\$\{([^}]+)\}$")


def _load(path):
    with open(path, "r", encoding="utf-8") as f:
        text = f.read()
    pkg = None
    for line in text.splitlines():
        if not line.strip():
            continue
        if not line.lstrip().startswith("#"):
            break
        m = _PKG.match(line.strip())
        if m:
            pkg = m.group(1)
            break
    data = yaml.safe_load(text) or {}
    if not isinstance(data, dict):
        raise ValueError(f"{path}: top level must be a mapping")
    return data, pkg


def _merge(dst, src, path=""):
    for k, v in src.items():
        here = f"{path}.{k}" if path else k
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v, here)
        else:
            m = _INTERP.match(v) if isinstance(v, str) else None
            if m and m.group(1) == here and k in dst:
                continue                                   # self-interpolation: keep the overridden value
            dst[k] = copy.deepcopy(v)
    return dst


def _place(root, package, content):
    if package in (None, "", "_global_"):
        return _merge(root, content)
    node = root
    for part in package.split("."):
        node = node.setdefault(part, {})
    _merge(node, content)
    return root


class _Composer:
    def __init__(self, cfg_dir, choices):
        self.dir = cfg_dir
        self.choices = choices            # group -> name overrides from the command line

    def _path(self, group, name):
        p = os.path.join(self.dir, group, name + ".yaml") if group else os.path.join(self.dir, name + ".yaml")
        if not os.path.exists(p):
            raise FileNotFoundError(f"config '{group + '/' if group else ''}{name}' not found under {self.dir}")
        return p

    def load(self, group, name, root):
        """Compose config `name` of `group` (''=root) into `root` at its package."""
        data, pkg = _load(self._path(group, name))
        defaults = data.pop("defaults", [])
        package = pkg if pkg is not None else (group.replace("/", ".") if group else "_global_")
        own_done = False
        for d in defaults:
            if d == "_self_":
                _place(root, package, data)
                own_done = True
            elif isinstance(d, str):
                if d.startswith("/"):
                    g, n = os.path.split(d[1:])
                else:
                    g, n = os.path.split(d)
                    g = os.path.join(group, g) if (group and g) else (g or group)
                self.load(g, n, root)
            elif isinstance(d, dict):
                for g, n in d.items():
                    g = g.lstrip("/")
                    n = self.choices.get(g, n)
                    if n is not None:
                        self.load(g, n, root)
            else:
                raise ValueError(f"unsupported defaults entry {d!r}")
        if not own_done:
            _place(root, package, data)      # Hydra >= 1.1: _self_ is implicitly last
        return root


def _parse_value(text):
    return yaml.safe_load(text)


def _resolve(node, root, path=(), trail=()):
    """Resolve `${a.b}` (absolute) and `${.x}` / `${..x}` (relative to the containing mapping) interpolations."""
    if isinstance(node, dict):
        return {k: _resolve(v, root, path + (k,), trail) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, path + (i,), trail) for i, v in enumerate(node)]
    if isinstance(node, str):
        m = _INTERP.match(node)
        if m:
            key = m.group(1)
            if key.startswith("."):
                ndots = len(key) - len(key.lstrip("."))
                base = path[:-1]
                base = base[:len(base) - (ndots - 1)] if ndots > 1 else base
                parts = base + tuple(key.lstrip(".").split("."))
            else:
                parts = tuple(key.split("."))
            if parts in trail:
                raise ValueError(f"circular interpolation through '{key}'")
            cur = root
            for part in parts:
                if isinstance(cur, list) and isinstance(part, int) and part < len(cur):
                    cur = cur[part]
                elif isinstance(cur, dict) and part in cur:
                    cur = cur[part]
                else:
                    raise ValueError(f"Config key '{key}' not found")          # config_utils.py:73-76
            return _resolve(cur, root, parts, trail + (parts,))
    return node


def compose(cfg_dir, config_name="config", overrides=()):
    """Compose `<cfg_dir>/<config_name>.yaml` with Hydra-style `overrides`; returns the resolved plain dict."""
    choices, sets = {}, []
    groups = {d for d in os.listdir(cfg_dir) if os.path.isdir(os.path.join(cfg_dir, d))}
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not key=value")
        key, val = ov.split("=", 1)
        key = key.lstrip("+")
        if key in groups and "." not in key:
            choices[key] = val
        else:
            sets.append((key, _parse_value(val)))
    root = _Composer(cfg_dir, choices).load("", config_name, {})
    for key, val in sets:
        node = root
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = val
    return _resolve(root, root)
