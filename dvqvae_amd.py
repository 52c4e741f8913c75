"""Import alias: the package directory is ``d-vqvae_amd/`` (the name the build contract fixes), which
is not a legal Python identifier.  ``import dvqvae_amd`` loads that directory as a package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "d-vqvae_amd")
_spec = importlib.util.spec_from_file_location(
    "dvqvae_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dvqvae_amd"] = _mod
_spec.loader.exec_module(_mod)
