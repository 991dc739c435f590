"""Minimal stand-ins for the Detectron2 plumbing the reference's hot-path modules are written
against (detectron2 is absent on both boxes): `Registry`, `ShapeSpec`, `configurable`.

Behavioural contract mirrored (boundary B1): classes register under their own
name in BACKBONE_REGISTRY / SEM_SEG_HEADS_REGISTRY / TRANSFORMER_DECODER_REGISTRY so that config files
resolve `MODEL.BACKBONE.NAME`, `MODEL.SEM_SEG_HEAD.{NAME,PIXEL_DECODER_NAME}` and
`MODEL.MASK_FORMER.TRANSFORMER_DECODER_NAME` (mask2former/modeling/pixel_decoder/fpn.py:21-33,
mask2former/modeling/transformer_decoder/maskformer_transformer_decoder.py:16-27);
`@configurable` lets a module be built either from a cfg (`Cls(cfg, ...)` -> `from_config`) or from
explicit keyword arguments.
"""
import functools
from dataclasses import dataclass
from typing import Optional


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._do_register(o.__name__, o)
                return o
            return deco
        self._do_register(obj.__name__, obj)
        return obj

    def _do_register(self, name, obj):
        if name in self._obj_map:
            raise KeyError(f"An object named '{name}' was already registered in '{self._name}' registry!")
        self._obj_map[name] = obj

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry! "
                           f"Known: {sorted(self._obj_map)}")
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map


BACKBONE_REGISTRY = Registry("BACKBONE")
SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")
TRANSFORMER_DECODER_REGISTRY = Registry("TRANSFORMER_MODULE")
META_ARCH_REGISTRY = Registry("META_ARCH")


@dataclass
class ShapeSpec:
    channels: Optional[int] = None
    height: Optional[int] = None
    width: Optional[int] = None
    stride: Optional[int] = None


def _called_with_cfg(*args, **kwargs):
    from .config import CfgNode
    if len(args) and isinstance(args[0], CfgNode):
        return True
    return isinstance(kwargs.get("cfg", None), CfgNode)


def configurable(init_func):
    """Decorate `__init__`: when the first argument is a CfgNode the class's `from_config(cfg, ...)`
    classmethod translates it into explicit keyword arguments."""
    assert init_func.__name__ == "__init__"

    @functools.wraps(init_func)
    def wrapped(self, *args, **kwargs):
        if _called_with_cfg(*args, **kwargs):
            from_config = type(self).from_config
            explicit = from_config(*args, **kwargs)
            init_func(self, **explicit)
        else:
            init_func(self, *args, **kwargs)
    return wrapped
