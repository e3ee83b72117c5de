"""Import shim: the package directory is named ``petal-decomposition_amd`` (with a hyphen, as the build
contract asks), which Python cannot import by name.  ``import petal_decomposition_amd`` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "petal-decomposition_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
