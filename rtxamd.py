"""Import shim: loads the package directory ``ray-tracing_amd/`` (not a valid identifier)
under the module name ``rtxamd_pkg`` and re-exports it."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "ray-tracing_amd")
if "rtxamd_pkg" not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        "rtxamd_pkg", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["rtxamd_pkg"] = _mod
    _spec.loader.exec_module(_mod)
_mod = sys.modules["rtxamd_pkg"]
globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("__")})
