"""Hierarchical yaml config with attribute access (the reference's utils/config.py surface:
`Config(load=True)` parses `--cfg FILE [--init_method URL] [KEY VAL ...]`, loads configs/pool/base.yaml,
then the `_BASE_RUN` / `_BASE_MODEL` / `_BASE` chain of FILE, then the command-line overrides
(utils/config.py:30-38,95-152,177-232); `Config(load=False, cfg_dict=...)` wraps a dict).

PyYAML reads exponents without a dot ("8e-6") as strings; like the reference (config.py:245-246) such
strings are coerced to float."""
import argparse
import copy
import os
import re

import yaml

_NUM = re.compile(r"^[+-]?\d+(\.\d*)?[eE][+-]?\d+$")


def _coerce(v):
    if isinstance(v, str) and _NUM.match(v):
        return float(v)
    if isinstance(v, dict):
        return {k: _coerce(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_coerce(x) for x in v]
    return v


def _merge(base, new):
    out = copy.deepcopy(base)
    for k, v in new.items():
        if k in ("_BASE", "_BASE_RUN", "_BASE_MODEL"):
            continue
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy(v)
    return out


def load_yaml_chain(path):
    """yaml at `path` merged over its _BASE_RUN, _BASE_MODEL and _BASE ancestors (relative paths)."""
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    here = os.path.dirname(os.path.abspath(path))
    base = {}
    for key in ("_BASE_RUN", "_BASE_MODEL", "_BASE"):
        if key in cfg:
            base = _merge(base, load_yaml_chain(os.path.normpath(os.path.join(here, cfg[key]))))
    return _merge(base, cfg)


def _find_pool_base(cfg_file):
    d = os.path.dirname(os.path.abspath(cfg_file))
    for _ in range(6):
        cand = os.path.join(d, "configs", "pool", "base.yaml")
        if os.path.exists(cand):
            return cand
        cand = os.path.join(d, "pool", "base.yaml")
        if os.path.exists(cand):
            return cand
        d = os.path.dirname(d)
    return None


class Config:
    def __init__(self, load=True, cfg_dict=None, cfg_level=None, argv=None):
        self._level = "cfg" + ("." + cfg_level if cfg_level is not None else "")
        if load:
            args = self._parse_args(argv)
            self.args = args
            cfg_dict = load_yaml_chain(args.cfg_file)
            base = _find_pool_base(args.cfg_file)
            if base is not None:
                with open(base) as f:
                    cfg_dict = _merge(yaml.safe_load(f) or {}, cfg_dict)
            cfg_dict = self._apply_overrides(cfg_dict, args.opts)
            cfg_dict = _coerce(cfg_dict)
            self.cfg_dict = cfg_dict
        self._update_dict(cfg_dict or {})

    @staticmethod
    def _parse_args(argv):
        p = argparse.ArgumentParser(description="dist_amd: DiST fine-tuning on MI355X")
        p.add_argument("--cfg", dest="cfg_file", required=True, help="path to the yaml config")
        p.add_argument("--init_method", default="tcp://127.0.0.1:9999", type=str)
        p.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="KEY VAL overrides, KEY = A.B.C")
        return p.parse_args(argv)

    @staticmethod
    def _apply_overrides(cfg, opts):
        opts = list(opts or [])
        assert len(opts) % 2 == 0, "overrides come as KEY VAL pairs"
        for k, v in zip(opts[0::2], opts[1::2]):
            node = cfg
            keys = k.split(".")
            for kk in keys[:-1]:
                node = node.setdefault(kk, {})
            node[keys[-1]] = yaml.safe_load(v)
        return cfg

    def _update_dict(self, d):
        for k, v in d.items():
            if isinstance(v, dict):
                v = Config(load=False, cfg_dict=v, cfg_level=k)
            else:
                v = _coerce(v)
            self.__dict__[k] = v

    def get_args(self):
        return self.args

    def deep_copy(self):
        return copy.deepcopy(self)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Config) else v) for k, v in self.__dict__.items()
                if not k.startswith("_") and k not in ("args", "cfg_dict")}

    def __repr__(self):
        return f"Config({self.to_dict()})"
