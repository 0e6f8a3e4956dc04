"""Imports the package directory `raytracer-rs_amd/` (not a valid Python identifier) as `raytracer_rs_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load_package():
    name = "raytracer_rs_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(_ROOT, "raytracer-rs_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    try:
        spec.loader.exec_module(mod)
    except Exception:
        del sys.modules[name]
        raise
    return mod
