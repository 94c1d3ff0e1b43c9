"""ConfigMixin / register_to_config: constructor arguments (with defaults) become `self.config.<name>`."""
import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


class ConfigMixin:
    config_name = None

    def register_to_config(self, **kwargs):
        d = dict(getattr(self, "_internal_dict", {}))
        d.update(kwargs)
        self._internal_dict = FrozenDict(d)

    @property
    def config(self):
        return self._internal_dict


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        params = [p for n, p in sig.parameters.items() if n != "self"]
        cfg = {p.name: p.default for p in params if p.default is not inspect.Parameter.empty}
        for p, a in zip(params, args):
            cfg[p.name] = a
        cfg.update({k: v for k, v in kwargs.items() if not k.startswith("_")})
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)
    return inner
