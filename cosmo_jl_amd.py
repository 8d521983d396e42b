"""Import shim: the product package lives in the directory `cosmo.jl_amd/` (the name the build contract asks for),
which Python cannot import by name because of the dot.  `import cosmo_jl_amd` loads that directory as a package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cosmo.jl_amd")
_spec = importlib.util.spec_from_file_location("cosmo_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cosmo_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
