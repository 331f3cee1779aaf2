"""Import alias: the package directory is `combo-avs_amd/` (not a valid Python identifier), so
`import combo_avs_amd` resolves here and re-exports that directory as a regular package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "combo-avs_amd")
_spec = importlib.util.spec_from_file_location(
    "combo_avs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["combo_avs_amd"] = _mod
_spec.loader.exec_module(_mod)
