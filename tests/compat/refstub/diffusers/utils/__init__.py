import logging as _pylog
from collections import OrderedDict
from dataclasses import fields

import torch

USE_PEFT_BACKEND = False          # diffusers 0.25.0 without `peft` installed (environment.yaml has no peft)


class BaseOutput(OrderedDict):
    """Dataclass outputs that also index like tuples (`out[0]`, `out.sample`)."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                OrderedDict.__setitem__(self, f.name, v)

    def __getitem__(self, k):
        if isinstance(k, str):
            return OrderedDict.__getitem__(self, k)
        return tuple(self.values())[k]

    def to_tuple(self):
        return tuple(self.values())


def deprecate(*args, **kwargs):
    return None


def is_torch_version(op, version):
    from packaging import version as V
    cur = V.parse(torch.__version__.split("+")[0])
    return {">=": cur >= V.parse(version), ">": cur > V.parse(version), "<": cur < V.parse(version),
            "<=": cur <= V.parse(version), "==": cur == V.parse(version)}[op]


def is_invisible_watermark_available():
    return False


def is_torch_xla_available():
    return False


def replace_example_docstring(example_docstring):
    def deco(fn):
        return fn
    return deco


def scale_lora_layers(model, weight):
    return None


def unscale_lora_layers(model, weight=None):
    return None


class _Logging:
    @staticmethod
    def get_logger(name=None):
        return _pylog.getLogger(name)


logging = _Logging()


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    from .. import _placeholder
    return _placeholder(name)
