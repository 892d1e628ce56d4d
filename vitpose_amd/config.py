"""Reader for the reference's Python config files (the mmcv `Config.fromfile` format the
ViTPose configs are written in, e.g.
configs/body/2d_kpt_sview_rgb_img/topdown_heatmap/coco/ViTPose_base_coco_256x192.py):
a config is a Python file whose top-level variables are the settings; `_base_` names one
or more files to inherit from (dicts merge key by key, `_delete_=True` replaces), and
`{{_base_.name}}` inside the text refers to a variable of the merged bases.

Only the `model=dict(...)` block feeds the hot path; the rest is returned untouched.
"""
import copy
import os
import re
import types

_BASE_KEY = '_base_'
_DELETE_KEY = '_delete_'
_VAR = re.compile(r'\{\{\s*_base_\.([\w\.]+)\s*\}\}')


class ConfigDict(dict):
    """dict with attribute access (cfg.model.backbone.embed_dim)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'") from None

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _merge(child, base):
    """Child values override; dicts merge recursively unless the child dict carries `_delete_`."""
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get(_DELETE_KEY, False):
            out[k] = _merge(v, out[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != _DELETE_KEY}
            out[k] = copy.deepcopy(v)
    return out


def _lookup(cfg, dotted):
    cur = cfg
    for part in dotted.split('.'):
        cur = cur[part]
    return cur


def _substitute(v, table):
    """Replace the placeholder strings of `{{_base_.x}}` references by their values."""
    if isinstance(v, str):
        if v in table:
            return copy.deepcopy(table[v])
        for ph, val in table.items():
            if ph in v:
                v = v.replace(ph, str(val))
        return v
    if isinstance(v, dict):
        return {k: _substitute(x, table) for k, x in v.items()}
    if isinstance(v, list):
        return [_substitute(x, table) for x in v]
    if isinstance(v, tuple):
        return tuple(_substitute(x, table) for x in v)
    return v


def _load(path):
    path = os.path.abspath(os.path.expanduser(path))
    if not os.path.isfile(path):
        raise FileNotFoundError(f'config file {path} does not exist')
    text = open(path, encoding='utf-8').read()
    refs = {}

    def repl(m):
        ph = f'_VPCFG_BASEVAR_{len(refs)}_'
        refs[ph] = m.group(1)
        return f'"{ph}"'
    # a quoted reference ("{{_base_.x}}") and a bare one both become one placeholder string
    text = re.sub(r'["\']' + _VAR.pattern + r'["\']', repl, text)
    text = _VAR.sub(repl, text)
    ns = {'__file__': path, '__name__': '_vitpose_cfg_'}
    exec(compile(text, path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    bases = cfg.pop(_BASE_KEY, None)
    if bases is not None:
        merged = {}
        for b in ([bases] if isinstance(bases, str) else bases):
            bcfg = _load(os.path.join(os.path.dirname(path), b))
            dup = set(merged) & set(bcfg)
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {sorted(dup)}')
            merged.update(bcfg)
        table = {ph: _lookup(merged, dotted) for ph, dotted in refs.items()}
        cfg = _merge(_substitute(cfg, table), merged)
    return cfg


class Config(ConfigDict):
    @staticmethod
    def fromfile(filename):
        return Config(_wrap(_load(filename)))
