"""Import shim: the package directory is `quoptimalcontrol.jl_amd/` (a dot is not legal in a
Python module name), so this module loads it under the importable alias
`quoptimalcontrol_jl_amd`.  `import quoptimalcontrol_jl_amd as qoc` with the repo root on
sys.path gives the package itself."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quoptimalcontrol.jl_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
