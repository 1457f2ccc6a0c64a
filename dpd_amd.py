"""Import shim: the package directory is named after the reference repository (`6g-pa-gan-dpd_amd`),
which is not a valid Python identifier; `import dpd_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "6g-pa-gan-dpd_amd")
_spec = importlib.util.spec_from_file_location(
    "dpd_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dpd_amd"] = _mod
_spec.loader.exec_module(_mod)
