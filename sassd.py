"""Import shim: the product package lives in the directory `sa-ssd_amd/` (not a valid Python identifier);
`import sassd` loads it under the module name `sassd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sa-ssd_amd")
_spec = importlib.util.spec_from_file_location("sassd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sassd"] = _mod
_spec.loader.exec_module(_mod)
