"""TEST-ONLY stand-in for the `diffusers==0.25.0` names the reference's hot-path modules import (environment.yaml:21).

diffusers is not installed in this image (no wheel, no network), so `/root/reference/src/*.py` and
`ip_adapter/attention_processor.py` cannot be imported as they are.  This package provides exactly the third-party
surface those files touch, written from the published semantics of that release, so that
`oracle/make_golden.py` and `tests/test_oracle.py` can EXECUTE THE REFERENCE'S OWN FIRST-PARTY CODE (UNet forward, block
sequencing, hacked BasicTransformerBlock / Transformer2DModel, attention processors) and pin `oracle/` to it.

It lives under tests/ and is never importable from the product: nothing in idm-vton_amd/, src/ or ip_adapter/ puts
tests/compat/refstub on sys.path.  Names the path never executes resolve to inert placeholder classes (see _Placeholder).
"""
import importlib.abc
import importlib.machinery
import sys
import types


class _Placeholder:
    """Inert class for names that are imported but never executed on the try-on path."""

    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__}: not on the IDM-VTON hot path (tests/compat/refstub placeholder)")


def _placeholder(name):
    return type(name, (_Placeholder,), {})


def _module_getattr(modname):
    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = _placeholder(name)
        setattr(sys.modules[modname], name, cls)
        return cls
    return __getattr__


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """`import diffusers.<anything not written out here>` -> an empty module of placeholders."""

    def find_spec(self, fullname, path, target=None):
        if fullname.startswith("diffusers.") and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        m.__getattr__ = _module_getattr(spec.name)
        return m

    def exec_module(self, module):
        pass


sys.meta_path.append(_StubFinder())           # appended: real files in this package win
__getattr__ = _module_getattr(__name__)
__version__ = "0.25.0"
