"""Import shim: the package directory is named `gan-inpainting_amd/` (not a Python identifier);
`import gan_inpainting_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gan-inpainting_amd")
_spec = importlib.util.spec_from_file_location(
    "gan_inpainting_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gan_inpainting_amd"] = _mod
_spec.loader.exec_module(_mod)
