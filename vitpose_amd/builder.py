"""Registry surface of the reference (mmpose/models/builder.py:6-39): ONE registry object
shared by BACKBONES / NECKS / HEADS / LOSSES / POSENETS, classes registered under their class
name, `build_*` = `cls(**cfg_without_type)`.

When the real mmpose is importable, `register_into_mmpose()` registers the HIP-backed classes
into its registry with force=True so that unchanged reference configs build them.
"""
import inspect


class Registry:
    """mmcv.utils.Registry-shaped (register_module decorator / module=..., build, get)."""

    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        if key not in self._module_dict:
            from . import models  # noqa: F401  (registers the HIP-backed classes on first use)
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f'module must be a class, but got {type(cls)}')
        for n in ([name] if isinstance(name, str) else (name or [cls.__name__])):
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self._name}')
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if not isinstance(force, bool):
            raise TypeError(f'force must be a boolean, but got {type(force)}')
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, default_args=None):
        """mmcv build_from_cfg semantics."""
        if not isinstance(cfg, dict):
            raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
        if 'type' not in cfg and not (default_args and 'type' in default_args):
            raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        obj_type = args.pop('type')
        if isinstance(obj_type, str):
            cls = self.get(obj_type)
            if cls is None:
                raise KeyError(f'{obj_type} is not in the {self._name} registry')
        elif inspect.isclass(obj_type):
            cls = obj_type
        else:
            raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
        return cls(**args)


MODELS = Registry('models')
BACKBONES = MODELS
NECKS = MODELS
HEADS = MODELS
LOSSES = MODELS
POSENETS = MODELS


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    """Raises TypeError for cfg=None, as the reference's registry build does
    (tests/test_models/test_top_down_head.py:149-151)."""
    return LOSSES.build(cfg)


def build_posenet(cfg):
    return POSENETS.build(cfg)


def register_into_mmpose():
    """Register the HIP-backed classes over mmpose's own (force=True).  Returns False when
    mmpose / mmcv are not importable (then use this module's registry)."""
    try:
        from mmpose.models.builder import MODELS as MM  # noqa: N811
    except Exception:
        return False
    for name, cls in MODELS.module_dict.items():
        MM.register_module(name=name, force=True, module=cls)
    return True
