"""Import alias: `import dib_amd` loads the package that lives in the (non-identifier) directory
`distributed-information-bottleneck.github.io_amd/`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "distributed-information-bottleneck.github.io_amd")
_spec = importlib.util.spec_from_file_location("dib_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dib_amd"] = _mod
_spec.loader.exec_module(_mod)
