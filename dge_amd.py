"""Import shim: the package directory is named `deep-gan-encoders_amd/` (not a valid Python
identifier), so `import dge_amd` loads it from there."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "deep-gan-encoders_amd")
_spec = importlib.util.spec_from_file_location("dge_amd", os.path.join(_d, "__init__.py"),
                                               submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dge_amd"] = _mod
_spec.loader.exec_module(_mod)
