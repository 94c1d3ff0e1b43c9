class KarrasDiffusionSchedulers:
    pass


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    from .. import _placeholder
    return _placeholder(name)
