"""Import shim: the package directory is named `idm-vton_amd/` (not a valid Python identifier), so this module loads it
under the importable name `idm_vton_amd`.  `import idm_vton_amd` / `from idm_vton_amd import ops` work as usual."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "idm-vton_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
