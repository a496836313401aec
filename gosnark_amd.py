"""Import shim: the package directory is named `go-snark-study_amd` (the repository's naming
contract), which is not an importable identifier.  `import gosnark_amd` loads that directory
as the package `gosnark_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "go-snark-study_amd")
_spec = importlib.util.spec_from_file_location(
    "gosnark_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gosnark_amd"] = _mod
_spec.loader.exec_module(_mod)
