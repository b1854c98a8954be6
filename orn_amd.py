"""Import alias: `import orn_amd` loads the package directory
`boosting-neural-video-representation-via-online-structural-reparameteration_amd/` (whose name is
not a valid Python identifier) under the module name `orn_amd`."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    'boosting-neural-video-representation-via-online-structural-reparameteration_amd')
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_DIR, '__init__.py'),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
