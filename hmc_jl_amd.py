"""Import shim: the package directory is `hmc.jl_amd/` (a dot in the name cannot be
imported directly), so `import hmc_jl_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hmc.jl_amd")
_spec = importlib.util.spec_from_file_location("hmc_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hmc_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
