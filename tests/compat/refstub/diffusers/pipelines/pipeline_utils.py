"""DiffusionPipeline: the handful of base-class services `StableDiffusionXLInpaintPipeline.__call__` uses."""
import contextlib

import torch

from ..configuration_utils import ConfigMixin


class DiffusionPipeline(ConfigMixin):
    _optional_components = []

    def __init__(self):
        self._internal_dict = {}

    def register_modules(self, **kwargs):
        for name, module in kwargs.items():
            setattr(self, name, module)

    @property
    def _execution_device(self):
        for name in ("unet", "vae"):
            m = getattr(self, name, None)
            if isinstance(m, torch.nn.Module):
                return next(m.parameters()).device
        return torch.device("cpu")

    @property
    def device(self):
        return self._execution_device

    def to(self, *args, **kwargs):
        for v in vars(self).values():
            if isinstance(v, torch.nn.Module):
                v.to(*args, **kwargs)
        return self

    @contextlib.contextmanager
    def progress_bar(self, iterable=None, total=None):
        class _Bar:
            def update(self, n=1):
                pass
        yield _Bar()

    def maybe_free_model_hooks(self):
        pass


class StableDiffusionMixin:
    pass
