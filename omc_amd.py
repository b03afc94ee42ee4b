"""Import shim: the package directory is named `optimalmatrixcompletion.jl_amd` (a dot cannot appear in an
`import` statement), so it is loaded here under the module name `omc_amd_pkg` and re-exported."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "optimalmatrixcompletion.jl_amd")
if "omc_amd_pkg" not in sys.modules:
    _spec = importlib.util.spec_from_file_location("omc_amd_pkg", os.path.join(_dir, "__init__.py"),
                                                   submodule_search_locations=[_dir])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["omc_amd_pkg"] = _mod
    _spec.loader.exec_module(_mod)
pkg = sys.modules["omc_amd_pkg"]
from omc_amd_pkg import *  # noqa: F401,F403,E402
from omc_amd_pkg import Engine, default_params, load, LIB_PATH, EXPORTS, OmcError  # noqa: F401,E402
